#!/usr/bin/env python3
"""What one rank of an N-GPU row-sharded run computes per step, on ONE GPU and without any
communication, for both scaling modes of bench.py:

  weak   : its own 10 000 queries through the encoder, pass 1 and pass 2 for ALL N*10 000 queries
           against its 100 000/N bank rows (in the chunks ShardedRange uses), finalize of its slice
  strong : BASELINE's 10 000-query batch in total: its own 10 000/N queries through the encoder,
           pass 1 and pass 2 for all 10 000 queries against its 100 000/N rows, finalize of its slice

The time of this against the single-GPU step bounds the scaling efficiency from above (the
exchange - all-gather of 1 040 B per query, all-reduce of 8 B per query, all-to-all of 4 KB per
query - comes on top; DESIGN.md section 6 prices it).
--layouts W: the strong mode for every 2-D layout R x Q of W ranks (range_amd.dist.make_layout: the bank
row-sharded over R ranks, Q = W / R such groups each serving its own queries): a rank holds 100 000 / R
rows, encodes 10 000 / W queries and scans the 10 000 / Q queries of its group.
--unchunked-pass1: one pass 1 over all scanned queries (the schedule before round 4), for A/B.
--cold: the protocol of rounds 2-4 (2 warm-up steps, 5 timed): a rank's 3 ms steps are then timed while
the chip is still raising its clock after idling (round 5, tools/clock_ramp.py: the first ~35 ms of
activity run up to 25 % slower) - the default now warms up for >= 150 ms and times 40 steps.
SHARD_P1_SPLITS=n: pass 1 of every chunk with n bank splits instead of the engine's choice (A/B).
Usage: python tools/shard_emulate.py [--json] [--cold] [--chunks k] [--layouts W] [--unchunked-pass1] [N ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from range_amd import _native, sh_table
from tools import synth
from range_amd.bank import prepare_bank

dev = torch.device("cuda:0")
w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
full = prepare_bank(*synth.make_bank(100000, 2024))
TABLE = sh_table.generate_table(40)      # load_model's default for an analytic checkpoint
as_json = "--json" in sys.argv
FORCE_CHUNKS = int(sys.argv[sys.argv.index("--chunks") + 1]) if "--chunks" in sys.argv else 0
if FORCE_CHUNKS:
    del sys.argv[sys.argv.index("--chunks"):sys.argv.index("--chunks") + 2]
LAYOUTS_W = int(sys.argv[sys.argv.index("--layouts") + 1]) if "--layouts" in sys.argv else 0
if LAYOUTS_W:
    del sys.argv[sys.argv.index("--layouts"):sys.argv.index("--layouts") + 2]
worlds = [int(v) for v in sys.argv[1:] if v.isdigit()] or [1, 2, 4, 8]
base = {}
# (mode, ranks in total, row shards R): a rank scans R x its own queries against 100 000 / R rows
cases = [(mode, W, W) for mode in ("strong", "weak") for W in worlds]
if LAYOUTS_W:
    cases = [("strong", 1, 1)] + [("strong", LAYOUTS_W, R) for R in (8, 4, 2, 1) if R <= LAYOUTS_W and LAYOUTS_W % R == 0]
for mode, WT, W in cases:                                       # W = ranks of a shard group from here on
    if True:
        n = 100000 // W
        B = 10000 // WT if mode == "strong" else 10000         # this rank's own queries
        eng = _native.HipEngine(dev)
        eng.set_encoder(40, 512, 2, 256, 0, [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                        [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]], sh_table=TABLE)
        eng.set_bank(full.keys[:n], full.values[:n], full.xyz[:n])
        x = torch.from_numpy(synth.make_queries(B, seed=7)).to(dev)
        e64, e32, xq = eng.encode(x)
        e32_all, xq_all = e32.repeat(W, 1).contiguous(), xq.repeat(W, 1).contiguous()
        # ShardedRange._chunk_bounds: 2 chunks from 1 024 queries per rank, 4 from 8 192 (--chunks k forces k)
        n_chunks = max(1, min(4 if B >= 8192 else 2, B // 512)) if W > 1 else 1
        if FORCE_CHUNKS:
            n_chunks = FORCE_CHUNKS
        cuts = [0] + [((B * c) // n_chunks + 32) // 64 * 64 for c in range(1, n_chunks)] + [B]

        # chunk-major order of the scanned queries (ShardedRange._scan), pass 1 per chunk with the
        # same bank splits, the shards' statistics merged in rank order (here: W copies of the own)
        ns1 = eng.p1_splits(W * max(hi - lo for lo, hi in zip(cuts[:-1], cuts[1:])))
        if os.environ.get("SHARD_P1_SPLITS"):          # (A/B of the chunks' bank-split count)
            ns1 = int(os.environ["SHARD_P1_SPLITS"])
        UNCHUNKED = "--unchunked-pass1" in sys.argv

        def step():
            eng.encode(x)
            sts = []
            if UNCHUNKED:
                st_all = eng.scan_stats(e32_all, xq_all, 12.0, 40.0, keep_logits=True)
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                first, m = W * lo, W * (hi - lo)
                if UNCHUNKED:
                    sts.append(st_all[first:first + m])
                    continue
                st = eng.scan_stats_at(e32_all[first:first + m], xq_all[first:first + m], 12.0, 40.0,
                                       first, W * B, n_splits=ns1)
                if W > 1:      # (its l is 1/W of the group's sum here: the emulation only times the kernel)
                    st = eng.merge_stats(st.unsqueeze(0).expand(W, m, 4).contiguous())
                sts.append(st)
            outs = []
            for (lo, hi), st in zip(zip(cuts[:-1], cuts[1:]), sts):
                first, m = W * lo, W * (hi - lo)
                part = eng.attend_kept(first, xq_all[first:first + m], 12.0, 40.0, 0.5, st)
                outs.append(eng.finalize(part.reshape(W, hi - lo, 1024), e64[lo:hi].contiguous()))
            return outs

        COLD = "--cold" in sys.argv
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if COLD:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            n_steps = 5
        else:
            # steady state: the chip raises its clock over the first ~35 ms of activity after idling
            a.record()
            step()
            b.record()
            b.synchronize()
            for _ in range(int(150.0 / max(a.elapsed_time(b), 0.1)) + 1):
                step()
            n_steps = 40
        eng.profile_enable(True)
        a.record()
        for _ in range(n_steps):
            step()
        b.record()
        b.synchronize()
        ms = a.elapsed_time(b) / n_steps
        k = {nm: round(eng.profile_read(i)[0] / n_steps, 3) for i, nm in enumerate(["encoder", "scan_stats", "attend"])}
        eng.profile_enable(False)
        qt, ns = eng.last_geometry()
        if WT == 1:
            base[mode] = ms
        total_q = B * WT
        rec = {"mode": mode, "n_gpus": WT, "layout": f"{W}x{WT // W} (row shards x query groups)",
               "ms_per_step_and_rank": round(ms, 3), "kernels_ms": k,
               "queries_total": total_q, "compute_only_geo_embeddings_per_s": round(total_q / ms * 1e3),
               "compute_only_speedup_vs_1": round((total_q / ms) / (10000 / base.get(mode, ms)), 3),
               "chunks": n_chunks, "last_pass2_grid": [qt, ns]}
        print(json.dumps(rec) if as_json else
              f"{mode:6s} N={WT} layout {W}x{WT // W}: {ms:7.3f} ms per step and rank  kernels {k}  grid {qt} x {ns}  "
              f"compute-only speed-up x{rec['compute_only_speedup_vs_1']}", flush=True)
        del eng
