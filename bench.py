#!/usr/bin/env python3
"""Headline benchmark: geo-embeddings/sec of the RANGE+ forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (encode -> stats -> attend -> finalize, i.e.
``load_model('RANGE+', beta=0.5)(locs)`` up to the device-resident (B,1280) float64 result) over
one batch of 10 000 synthetic queries per GPU, against the synthetic ``range_db_large`` bank
(N = 100 000 rows, SatCLIP-L40 encoder with H = 512; shapes are assumptions).
Inputs are resident in HBM when the timed region starts; the final device->host copy of the
reference's numpy contract is NOT in the timed region (PCIe-inclusive rate: DESIGN.md).

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the bank is row-sharded,
each rank serves its own 10 000 queries against all shards (range_amd/dist.py) - per-GPU work is
constant, so the reported scaling is weak.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from range_amd import _native, synth          # noqa: E402
from range_amd.bank import prepare_bank       # noqa: E402

FLOP_PAIR_ATTEND = 2 * (256 + 3 + 1024)       # pass 2 recomputing the logits, per (query, bank row)
FLOP_PAIR_ATTEND_KEPT = 2 * (3 + 1024)        # pass 2 on the logits pass 1 kept: geo tile + w @ V
FLOP_PAIR_STATS = 2 * (256 + 3)               # pass 1
FLOP_PAIR_REFERENCE = 4614                    # the reference's arithmetic
BANK_ROW_BYTES = (256 + 1024 + 3) * 4         # 5132 B
PEAK_F32_MATRIX_TFLOPS = 157.3                # MI355X_MICROARCH.md, dense f32 MFMA
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--queries", type=int, default=10_000, help="queries per GPU per step")
    ap.add_argument("--bank", default="range_db_large", choices=sorted(synth.BANK_ROWS))
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--beta", type=float, default=0.5)
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the row-sharded (torch.distributed) path even with one rank "
                         "(rehearsal of the N>1 code path on a 1-GPU box)")
    ap.add_argument("--layout", default="row-sharded", choices=["row-sharded", "query-sharded"],
                    help="N>1: row-sharded bank with the RCCL exchange (north-star layout, default) or "
                         "the control: the whole bank on every GPU, each rank embeds its own queries, "
                         "no collective on the data path")
    ap.add_argument("--shard-chunks", type=int, default=0,
                    help="query chunks of the sharded forward (0 = library default: 4 when N>1)")
    ap.add_argument("--cpu-sample", type=int, default=6144,
                    help="queries of the same workload timed on the host for cpu_baseline (0=off)")
    return ap.parse_args()


def encoder_params(L, H):
    from range_amd.ckpt import EncoderParams
    w = synth.make_encoder_weights(L, H, 256, 2, 1234)
    return w, EncoderParams(L, H, 2, 256, "analytic",
                            [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                            [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])


def cpu_baseline(weights, L, bank_arrays, n_sample, model, beta):
    """The oracle (CPU restatement of the reference, torch CPU ops in the reference's order and
    dtypes) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import range_oracle as O     # checker / baseline only
    locs, vals, keys = bank_arrays
    obank = O.prep_bank(locs, vals, keys)
    q = synth.make_queries(n_sample, seed=7)
    O.forward(q[:64], weights, L, obank, model, beta)          # warm-up
    t0 = time.perf_counter()
    O.forward(q, weights, L, obank, model, beta, chunk=512)
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "geo-embeddings/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n_sample} of the 10000-query batch, same bank, chunks of 512, "
                      f"{dt:.1f} s, host has {os.cpu_count()} logical CPUs"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 "
                             f"--nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    # RANGE_DIST_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks
    # (ranks share devices, collectives are staged through the host; the timing means nothing)
    backend = os.environ.get("RANGE_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local if backend == "nccl" else local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    replicated = world > 1 and a.layout == "query-sharded"
    sharded = (world > 1 and not replicated) or a.force_sharded
    if world > 1 or sharded:
        import torch.distributed as dist
        from range_amd.dist import ShardedRange, init_from_env, shard_rows
        init_from_env(backend)

    L, H = 40, a.hidden
    N = synth.BANK_ROWS[a.bank]
    weights, enc = encoder_params(L, H)
    bank_arrays = synth.make_bank(N, 2024)
    bank = prepare_bank(*bank_arrays)
    eng = _native.HipEngine(dev)
    eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, enc.weights, enc.biases)
    if not sharded:
        eng.set_bank(bank.keys, bank.values, bank.xyz, 0)
        n_local = N
    else:
        r0, r1 = shard_rows(N, world, rank)
        sh = bank.rows(r0, r1)
        eng.set_bank(sh.keys, sh.values, sh.xyz, r0)
        n_local = r1 - r0
        model = ShardedRange(eng, "RANGE+", a.beta, n_chunks=a.shard_chunks or None)

    B = a.queries
    x = torch.from_numpy(synth.make_queries(B, seed=7 + rank)).to(dev)
    out = torch.empty((B, 1280), dtype=torch.float64, device=dev)

    def step():
        if not sharded:
            eng.forward(x, _native.MODEL_RANGE_PLUS, a.beta, out=out)
        else:
            out.copy_(model(x))

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    eng.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    att_ms, att_n = eng.profile_read(_native.PROF_ATTEND)
    st_ms, st_n = eng.profile_read(_native.PROF_SCAN_STATS)
    en_ms, en_n = eng.profile_read(_native.PROF_ENCODER)
    eng.profile_enable(False)
    assert att_n >= a.steps and att_n % a.steps == 0, (att_n, a.steps)   # sharded: one per chunk
    assert bool(torch.isfinite(out).all())

    if rank == 0:
        launches_per_step = att_n // a.steps
        # row-sharded: every rank attends all queries (in chunks); replicated bank: only its own
        q_per_launch = B * (world if sharded else 1) // launches_per_step
        att_avg_ms = att_ms / att_n
        # which pass 2 ran: on the logits kept by pass 1 (the default) or recomputing them
        kept = eng.kept_queries() > 0
        flops = q_per_launch * n_local * (FLOP_PAIR_ATTEND_KEPT if kept else FLOP_PAIR_ATTEND)
        achieved = flops / (att_avg_ms * 1e-3) / 1e12
        # algorithmic bytes of one launch: the bank columns it reads once, per-query operands and
        # output, and (kept variant) 4 B per (query, row) of logits read back
        if kept:
            alg_bytes = n_local * (1024 + 4) * 4 + q_per_launch * (32 + 4096) + q_per_launch * n_local * 4
        else:
            alg_bytes = n_local * BANK_ROW_BYTES + q_per_launch * (1040 + 4096)
        traffic = None
        pmc = os.path.join(REPO, "profiles", "attend_pmc.json")
        if not sharded and os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if ("stored" in j.get("kernel", "")) == kept:
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        qt, ns = eng.last_geometry()
        total_q = B * world * a.steps
        res = {
            "metric": "geo-embeddings/sec (10k-query batch, range_db_large)",
            "value": total_q / dt,
            "unit": "geo-embeddings/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"RANGE+ beta={a.beta}, SatCLIP-L40 encoder (H={H}, synthetic "
                                   f"weights), {a.bank} (synthetic, N={N}), {B} queries per GPU "
                                   "per step, device-resident in/out",
                       "bank_rows": N, "queries_per_gpu": B, "hidden": H,
                       "bank_layout": ("single GPU" if world == 1 and not sharded else
                                       f"row-sharded x{world}" if sharded else
                                       f"replicated x{world} (query-sharded control)"),
                       "query_tiles": qt, "bank_splits": ns},
            "roofline": {"kernel": ("attend_stored_kernel<GEO> (pass 2 on kept logits: w@V, f32 MFMA)"
                                    if kept else
                                    "attend_kernel<GEO> (pass 2: logits + w@V, f32 MFMA)"),
                         "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MATRIX_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MATRIX_TFLOPS,
                         "traffic": traffic,
                         "avg_launch_ms": att_avg_ms, "launches": att_n,
                         "flop_per_launch": flops,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "kernels_ms_per_step": {"encoder": en_ms / a.steps, "scan_stats": st_ms / a.steps,
                                    "attend": att_ms / a.steps},
            "reference_equivalent_tflops": B * world * N * FLOP_PAIR_REFERENCE / (dt / a.steps) / 1e12,
        }
        if world == 1 and a.cpu_sample > 0:
            res["cpu_baseline"] = cpu_baseline(weights, L, bank_arrays, a.cpu_sample, "RANGE+", a.beta)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
