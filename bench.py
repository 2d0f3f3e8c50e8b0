#!/usr/bin/env python3
"""Headline benchmark: geo-embeddings/sec of the RANGE+ forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak]

One "step" = one pass of the hot path (encode -> stats -> attend -> finalize, i.e.
``load_model('RANGE+', beta=0.5)(locs)`` up to the device-resident (B,1280) float64 result) over
one batch of synthetic queries against the synthetic ``range_db_large`` bank (N = 100 000 rows,
SatCLIP-L40 encoder with H = 512; shapes are assumptions).  Inputs are resident in
HBM when the timed region starts; the device->host copy of the reference's numpy contract is NOT
in the timed region (it is measured separately and reported as ``value_host_contract``).

N > 1: one rank per GPU over RCCL, the bank row-sharded (range_amd/dist.py).  ``--gpus N`` without
WORLD_SIZE in the environment starts the N rank processes itself (fresh children of this process,
which has not touched a GPU) through ``torch.distributed.run`` and relays rank 0's JSON line;
launched by ``torch.distributed.run`` it is one of the ranks.

  --scaling strong (default) : BASELINE.json's metric - ONE batch of 10 000 queries in total,
                               10 000 / N per rank; for N > 1 a second, untimed-in-``value``
                               measurement of the weak mode follows and is reported under "weak"
  --scaling weak             : 10 000 queries per rank (per-GPU work constant)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BANKS = ("range_db_large", "range_db_med")

FLOP_PAIR_ATTEND = 2 * (256 + 3 + 1024)       # pass 2 recomputing the logits, per (query, bank row)
FLOP_PAIR_ATTEND_KEPT = 2 * (3 + 1024)        # pass 2 on the logits pass 1 kept: geo tile + w @ V
FLOP_PAIR_STATS = 2 * (256 + 3)               # pass 1
FLOP_PAIR_REFERENCE = 4614                    # the reference's arithmetic
BANK_ROW_BYTES = (256 + 1024 + 3) * 4         # 5132 B
KEY_ROW_BYTES = 256 * 4                       # keys-only scan (top-k side channel)
PEAK_F32_MATRIX_TFLOPS = 157.3                # MI355X_MICROARCH.md, dense f32 MFMA
PEAK_F64_MATRIX_TFLOPS = 78.6                 # dense f64 MFMA
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = --queries in total (BASELINE's 10k-query batch), "
                         "weak = --queries per GPU")
    ap.add_argument("--queries", type=int, default=10_000,
                    help="queries per step: in total (strong) or per GPU (weak)")
    ap.add_argument("--bank", default="range_db_large", choices=BANKS)
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--beta", type=float, default=0.5)
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the row-sharded (torch.distributed) path even with one rank "
                         "(rehearsal of the N>1 code path on a 1-GPU box)")
    ap.add_argument("--layout", default="row-sharded",
                    help="N>1: 'row-sharded' (north-star layout, default: the bank row-sharded over all N ranks, "
                         "RCCL exchange), 'query-sharded' (the control: the whole bank on every GPU, each rank "
                         "embeds its own queries, no collective on the data path) or 'RxQ' (2-D: the bank "
                         "row-sharded over groups of R ranks, Q = N/R groups each serving its own queries, "
                         "e.g. 2x4; Rx1 = row-sharded, 1xQ = query-sharded)")
    ap.add_argument("--shard-chunks", type=int, default=0,
                    help="query chunks of the sharded forward (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=4096,
                    help="queries of the same workload timed on the host for cpu_baseline (0=off)")
    ap.add_argument("--sweep", default=None, nargs="?", const="0,0.25,0.5,0.75,1",
                    help="BASELINE config 5: one step = the embeddings of the batch for EVERY beta of this "
                         "comma-separated list (default 0,0.25,0.5,0.75,1): pass 1 once, two passes 2 (the "
                         "semantic and the geographic retrieval), one blend + finalize per beta; the line's "
                         "value counts queries x betas")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (scan roofline, host-contract rate, weak-mode leg)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# parent: start the ranks (never touches a GPU)
# ------------------------------------------------------------------------------------------------
def spawn_ranks(a) -> int:
    """``python bench.py --gpus N`` from a plain shell: run N fresh rank processes under
    torch.distributed.run, pass their output through, return the launcher's exit code (non-zero
    when any rank failed: the elastic agent tears the others down)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    got_line = False
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        got_line = got_line or line.lstrip().startswith('{"metric"')
    rc = proc.wait()
    if rc == 0 and not got_line:
        print("bench.py: ranks exited 0 without a result line", file=sys.stderr)
        rc = 1
    return rc


# ------------------------------------------------------------------------------------------------
# rank process
# ------------------------------------------------------------------------------------------------
def encoder_params(L, H):
    from tools import synth
    from range_amd.ckpt import EncoderParams
    w = synth.make_encoder_weights(L, H, 256, 2, 1234)
    return w, EncoderParams(L, H, 2, 256, "analytic",
                            [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                            [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])


def note(msg: str) -> None:
    """Progress on stderr (stdout carries the one JSON line): the untimed extras of a default run take a few
    minutes in all - the CPU baseline alone half of that - and a silent process reads as a hung one."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cpus() -> int:
    """CPUs this process can actually USE: its affinity mask, capped by the cgroup's CPU quota (a GPU box
    of this pool shows 100+ logical CPUs and grants 16: a thread per visible CPU would spin its quota
    away at every OpenMP barrier).  Independent of --gpus and of OMP_NUM_THREADS."""
    import math
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            if parse:
                q, p = parse(open(path).read())
                if q == "max":
                    break
                q, p = int(q), int(p)
            else:
                q = int(open(path).read())
                p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q <= 0:
                    break
            n = min(n, max(1, math.ceil(q / p)))
            break
        except (OSError, ValueError):
            continue
    return n


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(weights, L, bank_arrays, n_sample, model, beta):
    """The oracle (CPU restatement of the reference, torch CPU ops in the reference's order and
    dtypes) timed on this box's host cores on a bounded sample of the same workload.

    ``value`` is the REFERENCE-SHAPED port: the spherical harmonics evaluated the way the reference
    evaluates them - L*L = 1600 functions per batch, each a fully expanded float64 polynomial whose
    powers are formed inside it (spherical_harmonics.py:35-42 over the generated file; 70 % of the
    reference's CPU time) - through ``oracle.sh_features_faithful``, which is pinned
    BITWISE to the reference's features (tests/test_oracle_golden.py).  ``port_recurrence`` is the
    same path with the stable three-term recurrence instead (what a sensible CPU port would run):
    reported beside it, never instead of it.  All threads with the SH / Siren / retrieval split, and
    one thread on a smaller sample."""
    import numpy as np
    import torch
    from oracle import range_oracle as O     # checker / baseline only
    from tools import synth
    locs, vals, keys = bank_arrays
    obank = O.prep_bank(locs, vals, keys)
    q = synth.make_queries(n_sample, seed=7, lat_max=90.0)
    table = O.load_ylm_table()

    def run(qs, chunk, faithful):
        t_sh = t_si = t_re = 0.0
        for i in range(0, qs.shape[0], chunk):
            ll = qs[i:i + chunk]
            t0 = time.perf_counter()
            y = O.sh_features_faithful(ll, table, L) if faithful else O.sh_features(ll, L, "analytic")
            t1 = time.perf_counter()
            e = torch.from_numpy(O.siren_forward(y, weights))
            e = (e / e.norm(p=2, dim=-1, keepdim=True)).numpy()
            t2 = time.perf_counter()
            O.retrieve(e, ll, obank, model, beta)
            t3 = time.perf_counter()
            t_sh += t1 - t0
            t_si += t2 - t1
            t_re += t3 - t2
        return t_sh, t_si, t_re

    # every CPU this process may run on, whatever OMP_NUM_THREADS the launcher exported (torchrun sets 1
    # for N > 1, this file's own launcher cpu_count / N): the baseline does not depend on --gpus
    n_prev = torch.get_num_threads()
    n_all = int(os.environ.get("RANGE_CPU_BASELINE_THREADS", "0")) or usable_cpus()
    torch.set_num_threads(n_all)
    host = (f"host: {cpu_model_name()}, {os.cpu_count()} logical CPUs, {n_all} usable by this process (affinity and "
            f"cgroup quota) = torch threads")

    def leg(faithful):
        run(q[:64], 64, faithful)                                  # warm-up
        sh, si, re_ = run(q, 512, faithful)
        dt = sh + si + re_
        n1 = max(32, min(192, n_sample // 16))
        torch.set_num_threads(1)
        try:
            sh1, si1, re1 = run(q[:n1], n1, faithful)
        finally:
            torch.set_num_threads(n_all)
        dt1 = sh1 + si1 + re1
        return {"value": n_sample / dt, "unit": "geo-embeddings/sec", "cores": n_all,
                "sample": f"{n_sample} of the 10000-query batch, same bank, chunks of 512, {dt:.1f} s; {host}",
                "split_s": {"sh_features": sh, "siren": si, "retrieval": re_},
                "split_q_per_s": {"sh_features": n_sample / sh, "siren": n_sample / si,
                                  "retrieval": n_sample / re_},
                "one_thread": {"value": n1 / dt1, "cores": 1, "sample": f"{n1} queries, {dt1:.1f} s",
                               "split_s": {"sh_features": sh1, "siren": si1, "retrieval": re1}}}

    res = leg(True)
    res["kind"] = "port"
    res["shape"] = ("reference-shaped: spherical harmonics as the reference evaluates them (1600 expanded float64 "
                    "polynomials per batch, one torch expression per (l,m); bitwise the reference's features), "
                    "SirenNet and retrieval in the reference's op order and dtypes")
    fast = leg(False)
    fast["kind"] = "port"
    fast["shape"] = "the same path with the spherical harmonics by the stable recurrence (not what the reference runs)"
    res["port_recurrence"] = fast
    torch.set_num_threads(n_prev)
    return res


def csrc_sha256():
    """SHA-256 over the sources of librange_hip.so (range_amd/_srchash.py: what build.sh embeds in the
    library and profiles/make_attend_pmc.py stamps the counter passes with)."""
    from range_amd._srchash import source_sha256
    return source_sha256()


def pmc_traffic(kernel, B, N, qt, ns):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/attend_pmc.json, written by tools/profile_bench.sh + profiles/make_attend_pmc.py: entries
    keyed by kernel and launch geometry and STAMPED with the SHA-256 of the kernel sources they were
    measured on).  Reported only while this checkout's sources hash to that stamp - counters of other
    kernels are not this run's traffic - else None with the reason.  Returns (bytes, source, entry)."""
    path = os.path.join(REPO, "profiles", "attend_pmc.json")
    try:
        entries = json.load(open(path))
    except Exception:
        return None, "no profiles/attend_pmc.json", None
    if isinstance(entries, dict):
        entries = [entries]
    sha = csrc_sha256()
    for e in entries:
        if (e.get("kernel_key") == kernel and e.get("queries") == B and e.get("bank_rows") == N
                and e.get("query_tiles") == qt and e.get("bank_splits") == ns):
            if e.get("csrc_sha256") != sha:
                return None, (f"profiles/attend_pmc.json was measured on other kernel sources (csrc sha256 "
                              f"{str(e.get('csrc_sha256'))[:12]}, this checkout {sha[:12]}): re-run tools/profile_bench.sh"), None
            return e.get("hbm_bytes_per_launch"), \
                f"profiles/attend_pmc.json (rocprofv3 --pmc passes of {e.get('source', '?')}; kernel sources sha256 {sha[:12]} = this checkout)", e
    return None, "no PMC entry for this workload/geometry in profiles/attend_pmc.json", None


#: untimed activity in front of the warm-up steps of every timed leg.  The chip lowers its clock when
#: idle and takes ~35 ms of continuous work to raise it again (tools/clock_ramp.py, round 5: the first
#: launches of a 2 ms kernel after a pause run up to 25 % slower, and with 3 ms of idle time between
#: launches the clock never recovers).  A rank's step of an 8-GPU run takes ~3 ms, so W = 3 warm-up steps
#: end inside that ramp; the steady state - what a job embedding millions of locations sees - is what
#: the metric is about.  The W warm-up steps and the K timed steps follow unchanged.
PREHEAT_MS = 150.0


def main():
    a = parse()
    backend = os.environ.get("RANGE_DIST_BACKEND", "nccl")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if backend == "threads":
            # rehearsal of the N-rank code path inside ONE process: the ranks are threads sharing cuda:0
            # (tools/thread_ranks.py; a box of this pool admits 6 processes on its card, the north star
            # has 8 ranks).  Every leg of the N-rank bench runs; the timing means nothing.
            from tools.thread_ranks import run_rank_threads, threaded_backend_available
            if not threaded_backend_available():
                raise SystemExit("RANGE_DIST_BACKEND=threads needs torch's in-process 'threaded' process group "
                                 "(torch.testing._internal.distributed.multi_threaded_pg): not in this torch build")
            import torch
            torch.cuda.init()             # (once, in front of the threads)
            res = run_rank_threads(a.gpus, lambda rank, world: rank_main(a, rank, 0, world, "threads"), timeout=3000.0)
            bad = {r: v for r, v in res.items() if v != "ok"}
            if bad:
                print(bad, file=sys.stderr)
            sys.exit(1 if bad else 0)
        sys.exit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from tools import rank_guard
    if (world > 1 and backend in ("nccl", "gloo") and not rank_guard.guarded()
            and os.environ.get("RANGE_BENCH_GUARD", "1") != "0"):
        # a rank of a multi-process job: THIS process (no torch, no GPU) only guards; the rank's work
        # runs in a fresh child whose stages it watches.  A first contact that fails or hangs on the
        # overlapped schedule is answered by a fresh child on the blocking one (tools/rank_guard.py) -
        # the run still yields a line, with dist.schedule = "blocking-fallback" - and whatever
        # happens, this rank ends well inside the driver's budget, non-zero when there is no line.
        dl = {k: float(os.environ[f"RANGE_BENCH_{k.upper()}_TIMEOUT_S"]) for k in rank_guard.DEFAULT_DEADLINES
              if os.environ.get(f"RANGE_BENCH_{k.upper()}_TIMEOUT_S")}
        sys.exit(rank_guard.guard_rank([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                       rank_guard.SHARDED_ATTEMPTS, dl,
                                       float(os.environ.get("RANGE_BENCH_TOTAL_TIMEOUT_S", "540"))))
    rank_main(a, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world, backend)


def rank_main(a, rank, local, world, backend):
    from tools.rank_guard import report_stage
    import numpy as np
    import torch          # (the first import of a fresh box takes 1-2 minutes: the guard's "start" stage, 240 s)

    from range_amd import _native
    from tools import synth
    from range_amd.bank import prepare_bank

    def fatal(msg):
        # (a configuration error: the rank guard does not answer it with another schedule - rank_guard.EX_FATAL)
        print(msg, file=sys.stderr, flush=True)
        raise SystemExit(78)

    if world != a.gpus:
        fatal(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        fatal("bench.py needs an MI355X; there is no CPU path")
    # RANGE_DIST_BACKEND=gloo / threads: rehearsal of the N>1 code path on a box with fewer GPUs than
    # ranks (ranks share devices; gloo stages the collectives through the host; the timing means nothing)
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        fatal(f"--gpus {world} over RCCL needs {world} GPUs, {n_dev} visible "
              "(RANGE_DIST_BACKEND=gloo or =threads rehearses the path on fewer)")
    dev = torch.device("cuda", local if backend == "nccl" else local % max(1, n_dev))
    torch.cuda.set_device(dev)
    dist = None
    import re as _re
    mm = _re.fullmatch(r"(\d+)x(\d+)", a.layout)
    if mm:
        row_shards = int(mm.group(1))
        if row_shards * int(mm.group(2)) != world:
            raise SystemExit(f"--layout {a.layout} does not multiply to --gpus {world}")
    elif a.layout == "query-sharded":
        row_shards = 1
    elif a.layout == "row-sharded":
        row_shards = world
    else:
        raise SystemExit(f"unknown --layout {a.layout!r}")
    replicated = world > 1 and row_shards == 1
    sharded = (world > 1 and not replicated) or a.force_sharded
    report_stage("init")  # (rendezvous + communicator: 180 s)
    if world > 1 or sharded:
        import torch.distributed as dist
        from range_amd.dist import ShardedRange, init_from_env, make_layout, shard_rows
        if backend != "threads":          # (the rank threads arrive inside an initialised group)
            # (timeout: 120 s instead of torch's 600 - RANGE_DIST_TIMEOUT_S; a second attempt of the
            # rank guard rendezvouses under its own store prefix)
            init_from_env(backend)
    report_stage("setup")

    L, H = 40, a.hidden
    N = synth.BANK_ROWS[a.bank]
    weights, enc = encoder_params(L, H)
    bank_arrays = synth.make_bank(N, 2024)
    bank = prepare_bank(*bank_arrays)
    eng = _native.HipEngine(dev)
    # the product's default for an 'analytic' checkpoint: the reference's generated SH polynomials
    from range_amd.range import sh_table_for
    table = sh_table_for(enc)
    eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, enc.weights, enc.biases, sh_table=table)
    model = None
    if not sharded:
        eng.set_bank(bank.keys, bank.values, bank.xyz, 0)
        n_local = N
    else:
        shard_group, shard_index, _ = make_layout(row_shards)
        r0, r1 = shard_rows(N, row_shards, shard_index)
        sh = bank.rows(r0, r1)
        eng.set_bank(sh.keys, sh.values, sh.xyz, r0)
        n_local = r1 - r0
        model = ShardedRange(eng, "RANGE+", a.beta, group=shard_group, n_chunks=a.shard_chunks or None)

    # ---- preflight of the sharded step (untimed): the FIRST steps over the job's collectives run under
    #      the rank guard's shortest deadline, the blocking schedule first (every collective waited for
    #      where it is issued: the form most likely to work on a backend the schedule has not met), then
    #      the overlapped one on the same batch; the two are bit-identical by design (dist.py), so
    #      torch.equal on the results is the check.  A mismatch on ANY rank sends every rank to the
    #      blocking schedule for the timed legs (dist.schedule = "blocking-fallback"); a hang or a
    #      failure ends this process - the guard then starts a fresh one with RANGE_DIST_BLOCKING=1,
    #      where only the blocking step is preflighted
    preflight = None
    if sharded:
        preflight = sharded_preflight(a, model, eng, dist, dev, rank, world, row_shards, shard_group, synth, torch)
        if preflight["schedule"] != "overlapped" and not model.blocking:
            model.blocking, model.pass1_chunked = True, False
            os.environ["RANGE_DIST_BLOCKING"] = "1"       # (the models of the later legs - 2-D layouts - too)
    report_stage("timed")

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def measure(scaling, steps, warmup, eng=eng, model=model, sharded=sharded):
        """One timed leg.  Returns a dict of raw measurements (rank-local except dt = max over ranks)."""
        if scaling == "strong":
            if a.queries % world:
                raise SystemExit(f"--queries {a.queries} is not a multiple of --gpus {world}")
            B = a.queries // world
        else:
            B = a.queries
        # the batch of the step: rank r owns queries [r*B, (r+1)*B) of ONE seeded batch
        # (pole to pole: the result does not depend on latitude, the parity check below does)
        q_all = synth.make_queries(B * world, seed=7, lat_max=90.0)
        q_host = q_all[rank * B:(rank + 1) * B]
        x = torch.from_numpy(q_host).to(dev)
        CH = 16384                                      # queries per engine call (LocationEncoder.chunk_size)
        if betas is None:
            out = torch.empty((B, 1280), dtype=torch.float64, device=dev)
        else:
            out = torch.empty((len(betas), B, 1280), dtype=torch.float64, device=dev)

        def step():
            if betas is None:
                if not sharded:
                    for lo in range(0, B, CH if B > 2 * CH else B):
                        hi = min(B, lo + (CH if B > 2 * CH else B))
                        eng.forward(x[lo:hi], _native.MODEL_RANGE_PLUS, a.beta, out=out[lo:hi])
                else:
                    model.embed(x, out=out, b_max=B)    # (one outer step up to 16 384 scanned queries)
            elif not sharded:
                # LocationEncoder.sweep: H and G once per chunk, one blend + finalize per beta
                for lo in range(0, B, CH):
                    e64, e32, xq = eng.encode(x[lo:lo + CH])
                    st = eng.scan_stats(e32, xq, 12.0, 40.0, keep_logits=True)
                    if eng.kept_queries() == e32.shape[0]:
                        Hs = eng.attend_kept(0, xq, 12.0, 40.0, 1.0, st)
                        Gs = eng.attend_kept(0, xq, 12.0, 40.0, 0.0, st)
                    else:
                        Hs = eng.attend(e32, xq, 12.0, 40.0, 1.0, st)
                        Gs = eng.attend(e32, xq, 12.0, 40.0, 0.0, st)
                    for j, b in enumerate(betas):
                        out[j, lo:lo + e64.shape[0]] = eng.finalize(eng.blend(Gs, Hs, b), e64)
            else:
                model.embed_sweep(x, betas, out=out, b_max=B)

        # untimed pre-heat (PREHEAT_MS): the same step until the chip has been busy long enough to hold
        # its clock; every rank runs the same count (the slowest rank's first step sets it)
        fence()
        t0 = time.perf_counter()
        step()
        fence()
        t_first = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([t_first], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            t_first = float(t.item())
        n_pre = min(200, int(PREHEAT_MS * 1e-3 / max(t_first, 1e-4)))
        for _ in range(n_pre):
            step()
        for _ in range(warmup):
            step()
        eng.profile_enable(True)
        if model is not None:
            model.comm_timing(True)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        prof = {k: eng.profile_read(v) for k, v in (("attend", _native.PROF_ATTEND),
                                                    ("scan_stats", _native.PROF_SCAN_STATS),
                                                    ("encoder", _native.PROF_ENCODER))}
        eng.profile_enable(False)
        comm_ms = model.comm_timing(False) if model is not None else None     # per kind of collective + "total"
        assert prof["attend"][1] >= steps and prof["attend"][1] % steps == 0, (prof, steps)
        assert bool(torch.isfinite(out[..., :8]).all()) and bool(torch.isfinite(out[..., -8:]).all())
        return {"B": B, "dt": dt, "prof": prof, "comm_ms": comm_ms, "q_host": q_host, "out": out,
                "kept": eng.kept_queries() > 0, "geometry": eng.last_geometry(), "preheat_steps": n_pre + 1}

    def per_step(comm_ms, steps):
        """Exposed communication per step, split per kind of collective (ShardedRange.comm_timing)."""
        if comm_ms is None:
            return None
        return {k: v / steps for k, v in comm_ms.items()}

    betas = None if a.sweep is None else [float(v) for v in a.sweep.split(",")]
    if rank == 0:
        note("timed leg")
    m = measure(a.scaling, a.steps, a.warmup)
    if rank == 0:
        note(f"timed leg done: {m['B'] * world * a.steps / m['dt']:,.0f} geo-embeddings/s; parity check and untimed extras follow")
    B, dt = m["B"], m["dt"]

    # ---- parity of the timed result (outside the timed region): 64 of rank 0's rows against the
    #      oracle in the reference's float32 op order and in exact float64
    parity = None
    parity_rows = None
    if rank == 0:
        from oracle import range_oracle as O     # checker only
        obank = O.prep_bank(*bank_arrays)
        idx = np.linspace(0, B - 1, num=min(64, B), dtype=np.int64)
        qs = m["q_host"][idx]
        sel = torch.from_numpy(idx).to(dev)
        got_all = (m["out"][sel] if betas is None else m["out"][:, sel]).cpu().numpy()
        got = got_all if betas is None else got_all[len(betas) // 2]
        beta_chk = a.beta if betas is None else betas[len(betas) // 2]
        # e-hat against the oracle fed with the same SH polynomials (CPU evaluation of the table;
        # queries run pole to pole, where the reference's polynomials are ill-conditioned and the
        # last bit of pow() shows: inside |lat| <= 45 the two agree to 1e-7), retrieval given e-hat
        e = O.encode(qs, weights, L, features=O.sh_features_faithful(qs, O.load_ylm_table(), L))
        band = np.abs(qs[:, 1]) <= 45.0
        got_e = got[:, 1024:]
        ref32 = O.retrieve(got_e, qs, obank, "RANGE+", beta_chk)
        ref64 = O.retrieve64(got_e, qs, obank, "RANGE+", beta_chk)
        parity = {"rows": int(idx.size),
                  "max_abs_vs_reference_f32_order": float(np.abs(got - ref32).max()),
                  "max_abs_vs_f64_oracle": float(np.abs(got[:, :1024] - ref64).max()),
                  "ehat_max_abs_lat_le_45": float(np.abs(got_e - e)[band].max()),
                  "ehat_max_abs_all_latitudes": float(np.abs(got_e - e).max())}
        if betas is not None:      # every beta of the sweep against the float64 oracle
            parity["sweep_max_abs_vs_f64_oracle"] = {
                str(b): float(np.abs(got_all[j][:, :1024] - O.retrieve64(got_e, qs, obank, "RANGE+", b)).max())
                for j, b in enumerate(betas)}
            if not max(parity["sweep_max_abs_vs_f64_oracle"].values()) < 1e-4:
                raise SystemExit(f"bench parity failed: {parity}")
        parity_rows = {"idx": idx, "exact": got[:, :1024].copy(), "ref64": ref64}
        if not parity["ehat_max_abs_lat_le_45"] < 1e-6:
            raise SystemExit(f"bench parity failed: {parity}")
        if not parity["max_abs_vs_reference_f32_order"] < 1e-4:
            raise SystemExit(f"bench parity failed: {parity}")

    # ---- untimed extras
    weak = None
    default_workload = betas is None and a.queries == 10_000
    if world > 1 and a.scaling == "strong" and not a.no_extras and default_workload:
        mw = measure("weak", a.steps, max(1, a.warmup))
        weak = {"value": mw["B"] * world * a.steps / mw["dt"], "ms_per_step": mw["dt"] / a.steps * 1e3,
                "queries_per_gpu": mw["B"], "comm_ms_exposed_per_step": per_step(mw["comm_ms"], a.steps)}
    control = None
    if world > 1 and sharded and a.scaling == "strong" and not a.no_extras and default_workload:
        # the control experiment of SURVEY.md 8(e): the same batch with the whole bank on every GPU,
        # each rank embedding its own queries end to end - no collective on the data path.  What the
        # row-sharded figure loses against it is the price of the exchange (and of scanning all
        # queries on every rank).
        engc = _native.HipEngine(dev)
        engc.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, enc.weights, enc.biases, sh_table=table)
        engc.set_bank(bank.keys, bank.values, bank.xyz, 0)
        Bc = a.queries // world
        xc = torch.from_numpy(synth.make_queries(Bc * world, seed=7, lat_max=90.0)[rank * Bc:(rank + 1) * Bc]).to(dev)
        outc = torch.empty((Bc, 1280), dtype=torch.float64, device=dev)
        for _ in range(max(1, a.warmup)):
            engc.forward(xc, _native.MODEL_RANGE_PLUS, a.beta, out=outc)
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            engc.forward(xc, _native.MODEL_RANGE_PLUS, a.beta, out=outc)
        fence()
        tc = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        # same queries, same bank: the two layouts must agree (split-order rounding only)
        agree = float((outc - m["out"]).abs().max())
        control = {"layout": "bank replicated, query-sharded, no collective", "value": Bc * world * a.steps / float(tc.item()),
                   "ms_per_step": float(tc.item()) / a.steps * 1e3, "max_abs_vs_row_sharded": agree}
        engc.close()
    layouts = None
    if world > 2 and sharded and row_shards == world and a.scaling == "strong" and not a.no_extras and default_workload:
        # the 2-D layouts R x Q of the same job (the bank row-sharded over groups of R ranks, Q groups each
        # serving its own queries: collectives span R ranks, shards are W/R times larger) in the same
        # invocation, next to the north-star layout W x 1 that `value` reports
        layouts = {}
        for R in [r for r in range(2, world) if world % r == 0]:
            g2, si2, _ = make_layout(R)
            r0, r1 = shard_rows(N, R, si2)
            e2 = _native.HipEngine(dev)
            e2.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, enc.weights, enc.biases, sh_table=table)
            sh2 = bank.rows(r0, r1)
            e2.set_bank(sh2.keys, sh2.values, sh2.xyz, r0)
            m2 = measure("strong", a.steps, max(1, a.warmup), eng=e2,
                         model=ShardedRange(e2, "RANGE+", a.beta, group=g2, n_chunks=a.shard_chunks or None), sharded=True)
            layouts[f"{R}x{world // R}"] = {
                "value": m2["B"] * world * a.steps / m2["dt"], "ms_per_step": m2["dt"] / a.steps * 1e3,
                "bank_rows_per_gpu": r1 - r0, "comm_ms_exposed_per_step": per_step(m2["comm_ms"], a.steps),
                "max_abs_vs_row_sharded": float((m2["out"] - m["out"]).abs().max())}
            del m2
            e2.close()
    scan = None
    with_topk = None
    host_contract = None
    opt_in = None
    envelope = None
    if world == 1 and not sharded and not a.no_extras and default_workload:
        note("extras: scan roofline")
        scan = scan_roofline(eng, synth, torch, dev, N, bank)
        note("extras: return_topk cost")
        with_topk = return_topk_cost(eng, synth, torch, dev, a)
        note("extras: host contract")
        host_contract = host_contract_rate(eng, synth, torch, dev, a.beta, a.queries)
        note("extras: opt-in bf16x3")
        opt_in = opt_in_bf16x3(eng, measure, parity_rows, a, torch, dev)
        note("extras: kept-logit envelope (a 10^6-row bank)")
        envelope = kept_logits_envelope(measure, a, torch, dev, enc, table, bank, N, L, H)

    if rank == 0:
        att_ms, att_n = m["prof"]["attend"]
        st_ms, st_n = m["prof"]["scan_stats"]
        en_ms, en_n = m["prof"]["encoder"]
        launches_per_step = att_n // a.steps
        # row-sharded: every rank attends all queries (in chunks); replicated bank: only its own
        q_scanned = B * (row_shards if sharded else 1)
        # (the sweep runs pass 2 twice per query: the semantic and the geographic retrieval)
        q_per_launch = q_scanned * (2 if betas is not None else 1) // launches_per_step
        att_avg_ms = att_ms / att_n
        kept = m["kept"]     # which pass 2 ran: on the logits pass 1 kept (default) or recomputing
        flops = q_per_launch * n_local * (FLOP_PAIR_ATTEND_KEPT if kept else FLOP_PAIR_ATTEND)
        achieved = flops / (att_avg_ms * 1e-3) / 1e12
        # algorithmic bytes of one launch: the bank columns it reads once, per-query operands and
        # output, and (kept variant) 4 B per (query, row) of logits read back
        if kept:
            alg_bytes = n_local * (1024 + 4) * 4 + q_per_launch * (32 + 4096) + q_per_launch * n_local * 4
        else:
            alg_bytes = n_local * BANK_ROW_BYTES + q_per_launch * (1040 + 4096)
        qt, ns = m["geometry"]
        kernel_key = "attend_stored_kernel<true>" if kept else "attend_kernel<true>"
        traffic, traffic_source, pmc_entry = (pmc_traffic(kernel_key, q_per_launch, n_local, qt, ns)
                                              if not sharded else (None, "not profiled for the sharded layout", None))
        step_s = dt / a.steps
        st_flops = q_scanned * n_local * FLOP_PAIR_STATS
        en_flops = B * 2 * (L * L * H + H * H + 256 * H)
        executed = (q_scanned * n_local * ((FLOP_PAIR_ATTEND_KEPT if kept else FLOP_PAIR_ATTEND) * (2 if betas else 1)
                                           + FLOP_PAIR_STATS) + en_flops)
        n_out = 1 if betas is None else len(betas)
        total_q = B * world * a.steps * n_out
        per_gpu = "in total" if a.scaling == "strong" else "per GPU"
        res = {
            "metric": ("geo-embeddings/sec (10k-query batch, range_db_large)" if default_workload and a.bank == "range_db_large"
                       else f"geo-embeddings/sec ({a.queries}-query batch, {a.bank}"
                            + (f", beta sweep x{n_out}: queries x betas per second)" if betas else ")")),
            "value": total_q / dt,
            "unit": "geo-embeddings/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            # what a timed leg is (for comparisons across rounds): version 1 = rounds 1-4 (W warm-up steps, K
            # timed steps); version 2 = round 5 on: PREHEAT_MS of untimed steps in front of the W warm-up steps
            "protocol": {"version": 2, "preheat_ms": PREHEAT_MS},
            "preheat": {"untimed_steps_before_warmup": m["preheat_steps"], "target_ms": PREHEAT_MS,
                        "why": "the chip needs ~35 ms of continuous work after idling to hold its clock (tools/clock_ramp.py)"},
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"RANGE+ {'beta sweep ' + a.sweep if betas else 'beta=' + str(a.beta)}, SatCLIP-L40 encoder (H={H}, synthetic "
                                   f"weights), {a.bank} (synthetic, N={N}), {a.queries} queries "
                                   f"{per_gpu} per step ({B} per GPU), device-resident in/out",
                       "bank_rows": N, "queries_total": B * world, "queries_per_gpu": B, "hidden": H,
                       "bank_layout": ("single GPU" if world == 1 and not sharded else
                                       (f"row-sharded x{world}" if row_shards == world else
                                        f"2-D: row-sharded x{row_shards}, {world // row_shards} query groups") if sharded else
                                       f"replicated x{world} (query-sharded control)"),
                       "bank_rows_per_gpu": n_local, "query_tiles": qt, "bank_splits": ns},
            "roofline": {"kernel": ("attend_stored_kernel<GEO> (pass 2 on kept logits: w@V, f32 MFMA)"
                                    if kept else
                                    "attend_kernel<GEO> (pass 2: logits + w@V, f32 MFMA)"),
                         "bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MATRIX_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MATRIX_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         # all kernels of the step (pass 1 writes the kept logits, pass 2 reads them back), and
                         # against SURVEY 8(d)'s HBM-minimal one bank pass + per-query I/O
                         "traffic_per_step": None if pmc_entry is None else pmc_entry.get("hbm_bytes_per_step"),
                         "traffic_per_step_over_hbm_minimal":
                             None if pmc_entry is None else pmc_entry.get("hbm_bytes_per_step_over_survey_minimal"),
                         # the logits pass 1 keeps for pass 2 (4 B per (query, bank row) of a call): what they
                         # cost, where keeping stops, and what the step makes without them
                         "kept_logits_bytes": (((q_per_launch + 63) // 64) * ((n_local + 15) // 16) * 4096) if kept else 0,
                         **({} if envelope is None else envelope),
                         "avg_launch_ms": att_avg_ms, "launches": att_n,
                         "queries_per_launch": q_per_launch,
                         "flop_per_launch": flops,
                         "algorithmic_bytes_per_launch": alg_bytes},
            # every kernel of the step against the peak that bounds it (HIP events on the launch
            # stream, averaged over the timed steps)
            "kernels": {
                "encoder": {"ms_per_step": en_ms / a.steps, "bound": "mfma f64",
                            "achieved_tflops": en_flops / (en_ms / a.steps * 1e-3) / 1e12,
                            "frac": en_flops / (en_ms / a.steps * 1e-3) / 1e12 / PEAK_F64_MATRIX_TFLOPS},
                "scan_stats": {"ms_per_step": st_ms / a.steps, "bound": "mfma f32",
                               "achieved_tflops": st_flops / (st_ms / a.steps * 1e-3) / 1e12,
                               "frac": st_flops / (st_ms / a.steps * 1e-3) / 1e12 / PEAK_F32_MATRIX_TFLOPS},
                "attend": {"ms_per_step": att_ms / a.steps, "bound": "mfma f32",
                           "achieved_tflops": achieved, "frac": achieved / PEAK_F32_MATRIX_TFLOPS},
                "other_ms_per_step": step_s * 1e3 - (en_ms + st_ms + att_ms) / a.steps},
            "executed_tflops": executed / step_s / 1e12,
            "reference_equivalent_tflops": B * world * n_out * N * FLOP_PAIR_REFERENCE / step_s / 1e12,
            "parity_max_abs": parity["max_abs_vs_reference_f32_order"],
            "parity": parity,
        }
        if dist is not None:
            from range_amd.dist import dist_timeout_s
            be = dist.get_backend()
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if be == "nccl" else None
            except Exception:  # noqa: BLE001
                rccl = None
            res["dist"] = {"backend": be, "world_size": dist.get_world_size(), "rccl_version": rccl,
                           "torch": torch.__version__, "timeout_s": dist_timeout_s(),
                           "layout": a.layout if world > 1 else "row-sharded (forced, one rank)",
                           "row_shards": row_shards if sharded else 1, "query_groups": world // row_shards if sharded else world,
                           # which schedule the timed legs ran: "overlapped" (the default), "blocking-fallback"
                           # (the preflight's verdict, or a second attempt of the rank guard), "blocking" (asked
                           # for with RANGE_DIST_BLOCKING=1)
                           "schedule": preflight["schedule"] if preflight else "none (query-sharded: no collective on the data path)",
                           "preflight": preflight,
                           "guard": {"guarded": bool(os.environ.get("RANGE_GUARD_FD")),
                                     "attempt": int(os.environ.get("RANGE_GUARD_ATTEMPT", "1")),
                                     "previous_failure": os.environ.get("RANGE_GUARD_PREVIOUS_FAILURE") or None},
                           "comm_ms_exposed_per_step": per_step(m["comm_ms"], a.steps)}
        if weak is not None:
            res["weak"] = weak
        if control is not None:
            res["control_query_sharded"] = control
        if layouts:
            res["layouts"] = layouts
            best = max(layouts, key=lambda k: layouts[k]["value"])
            res["best_2d_layout"] = {"layout": best, **layouts[best]}
        if scan is not None:
            res["roofline_scan"] = scan
        if with_topk is not None:
            res["return_topk"] = with_topk
        if host_contract is not None:
            res["value_host_contract"] = host_contract["value"]
            res["host_contract"] = host_contract
        if opt_in is not None:
            res["opt_in_bf16x3"] = opt_in
        if world == 1 and a.cpu_sample > 0:
            note(f"cpu_baseline: the oracle on {usable_cpus()} host threads, {a.cpu_sample} queries (two legs)")
            res["cpu_baseline"] = cpu_baseline(weights, L, bank_arrays, a.cpu_sample, "RANGE+", a.beta)
            note("cpu_baseline done")
        print(json.dumps(res), flush=True)
    report_stage("done")
    if dist is not None:
        dist.barrier()
        if backend != "threads":
            dist.destroy_process_group()


def sharded_preflight(a, model, eng, dist, dev, rank, world, row_shards, shard_group, synth, torch):
    """See the call site.  Returns the ``dist.preflight`` record (the same on every rank):
    {"schedule", "blocking_ms", "overlapped_ms", "bit_identical", "ranks_agree", "queries_per_rank"}."""
    from range_amd.dist import ShardedRange
    from tools.rank_guard import report_stage
    B = max(1, a.queries // world) if a.scaling == "strong" else a.queries
    B = min(B, 16384)
    q = synth.make_queries(B * world, seed=7, lat_max=90.0)[rank * B:(rank + 1) * B]
    x = torch.from_numpy(q).to(dev)
    fault = os.environ.get("RANGE_BENCH_INJECT", "")      # (tests: tests/test_gpu_tools.py)

    def run(m):
        out = torch.empty((B, 1280), dtype=torch.float64, device=dev)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        m.embed(x, out=out, b_max=B)
        torch.cuda.synchronize(dev)
        return out, (time.perf_counter() - t0) * 1e3

    def agree(flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    report_stage("preflight")
    rec = {"queries_per_rank": B, "chunks_overlapped": len(model._chunk_bounds(B))}
    asked_blocking = model.blocking          # RANGE_DIST_BLOCKING=1: by hand, or a second attempt of the guard
    mb = model if asked_blocking else ShardedRange(eng, "RANGE+", a.beta, group=shard_group, n_chunks=a.shard_chunks or None)
    if not asked_blocking:
        mb.blocking, mb.pass1_chunked = True, False
    out_b, rec["blocking_ms"] = run(mb)
    finite = bool(torch.isfinite(out_b).all())
    if asked_blocking:
        second = int(os.environ.get("RANGE_GUARD_ATTEMPT", "1")) > 1
        rec.update(schedule="blocking-fallback" if second else "blocking", overlapped_ms=None, bit_identical=None,
                   why=(os.environ.get("RANGE_GUARD_PREVIOUS_FAILURE") if second else "RANGE_DIST_BLOCKING=1"))
        if not agree(finite):
            raise SystemExit(f"bench preflight: the blocking schedule produced non-finite rows on some rank: {rec}")
        return rec
    report_stage("preflight")                # (a deadline of its own for the overlapped step)
    if fault == "hang_overlapped" and rank == world - 1:
        time.sleep(10_000)
    out_o, rec["overlapped_ms"] = run(model)
    same = bool(torch.equal(out_b, out_o))
    if fault == "mismatch_overlapped" and rank == world - 1:
        same = False
    rec["bit_identical"] = same
    rec["ranks_agree"] = agree(same and finite)
    rec["schedule"] = "overlapped" if rec["ranks_agree"] else "blocking-fallback"
    if not rec["ranks_agree"]:
        rec["why"] = "overlapped != blocking on some rank (or non-finite rows): the timed legs run the blocking schedule"
    return rec


def kept_logits_envelope(measure, a, torch, dev, enc, table, bank, N, L, H):
    """The envelope of the kept-logit scheme (pass 1 writes 4 B per (query, bank row), pass 2 reads
    them back instead of recomputing e . K^T: 25 % fewer MFMAs in pass 2 for 20x the HBM-minimal
    traffic of the step - the right trade while the step is MFMA-bound):
      keep_switch_off  a call keeps its logits while they fit in HALF of the free device memory
                       (include/range_hip.h: range_scan_stats); beyond, pass 2 recomputes
      recompute_path   the same step with keeping switched off (RANGE_KEEP_LOGITS=0), on this bank and
                       on a DRAM-sized one (N = 10^6 rows: 5.1 GB of bank, generated on the device),
                       where the kept logits of a 10 000-query step are 40 GB."""
    import numpy as np
    from range_amd import _native
    from range_amd.bank import PreparedBank
    free_b, total_b = torch.cuda.mem_get_info(dev)
    out = {"keep_switch_off": {"rule": "logits of a call are kept while 4 B x queries x bank rows <= half of the free device memory",
                               "free_bytes_now": int(free_b), "max_pairs_kept": int(free_b // 8),
                               "queries_per_call_at_this_bank": int(free_b // 8 // N),
                               "queries_per_call_at_1e6_rows": int(free_b // 8 // 1_000_000)}}

    def engine(keep, b):
        if not keep:
            os.environ["RANGE_KEEP_LOGITS"] = "0"          # read at range_create
        try:
            e = _native.HipEngine(dev)
        finally:
            os.environ.pop("RANGE_KEEP_LOGITS", None)
        e.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, enc.weights, enc.biases, sh_table=table)
        e.set_bank(b.keys, b.values, b.xyz, 0)
        return e

    steps = max(3, a.steps // 4)
    e0 = engine(False, bank)
    m = measure(a.scaling, steps, 1, eng=e0, model=None, sharded=False)
    rec = {"this_bank": {"bank_rows": N, "value": m["B"] * steps / m["dt"], "ms_per_step": m["dt"] / steps * 1e3,
                         "kept": bool(m["kept"])}}
    e0.close()
    # a DRAM-sized bank: the values (4.1 GB) no longer sit in the 256 MB Infinity Cache
    n_big = 1_000_000
    g = torch.Generator(device=dev).manual_seed(77)
    keys = torch.nn.functional.normalize(torch.randn((n_big, 256), generator=g, device=dev), dim=1)
    vals = torch.randn((n_big, 1024), generator=g, device=dev)
    xyz = torch.nn.functional.normalize(torch.randn((n_big, 3), generator=g, device=dev), dim=1)
    big = PreparedBank(keys.cpu().numpy(), vals.cpu().numpy(), xyz.cpu().numpy())
    del keys, vals, xyz
    for keep in (True, False):
        e1 = engine(keep, big)
        m = measure(a.scaling, steps, 1, eng=e1, model=None, sharded=False)
        rec[f"rows_1e6_{'kept' if keep else 'recompute'}"] = {
            "bank_rows": n_big, "value": m["B"] * steps / m["dt"], "ms_per_step": m["dt"] / steps * 1e3,
            "kept": bool(m["kept"]), "kept_logits_bytes": int(((m["B"] + 63) // 64) * ((n_big + 15) // 16) * 4096) if m["kept"] else 0}
        e1.close()
    out["recompute_path"] = rec
    return out


def opt_in_bf16x3(eng, measure, parity_rows, a, torch, dev):
    """The same step with the OPT-IN arithmetic of pass 2 (load_model(..., pv_mode="bf16x3"):
    w @ V on three bf16 planes of both operands).  Never `value`: the headline is the exact float32
    path.  Reported with its error on the rows of the parity check: against the float64 oracle
    (next to the exact kernel's own error there) and against the exact kernel's output."""
    from range_amd import _native
    import numpy as np
    eng.set_pv_mode("bf16x3")
    try:
        m = measure(a.scaling, max(3, a.steps // 4), 2)
    finally:
        eng.set_pv_mode("exact")
    steps = max(3, a.steps // 4)
    got = m["out"][torch.from_numpy(parity_rows["idx"]).to(dev)].cpu().numpy()[:, :1024]
    att_ms, att_n = m["prof"]["attend"]
    return {"what": "opt-in pv_mode='bf16x3' (never the default, never `value`)",
            "value": m["B"] * steps / m["dt"], "unit": "geo-embeddings/s",
            "ms_per_step": m["dt"] / steps * 1e3, "pass2_ms": att_ms / att_n,
            "bf16_tflops": m["B"] * eng.n_rows * 1024 * 2 * 6 / (att_ms / att_n * 1e-3) / 1e12,
            "frac_of_bf16_peak": m["B"] * eng.n_rows * 1024 * 2 * 6 / (att_ms / att_n * 1e-3) / 1e12 / 2500.0,
            "max_abs_vs_f64_oracle": float(np.abs(got - parity_rows["ref64"]).max()),
            "exact_kernel_max_abs_vs_f64_oracle": float(np.abs(parity_rows["exact"] - parity_rows["ref64"]).max()),
            "max_abs_vs_exact_kernel": float(np.abs(got - parity_rows["exact"]).max())}


def scan_rocprof(n_rows, nq, keys_mode):
    """The rocprofv3 median of the same call from the committed kernel-trace passes
    (profiles/scan_summary.json: tools/profile_bench.sh + profiles/scan_summarize.py, stamped with the
    SHA-256 of the kernel sources measured) - reported only while this checkout's sources hash to that
    stamp, like ``pmc_traffic``."""
    path = os.path.join(REPO, "profiles", "scan_summary.json")
    try:
        d = json.load(open(path))
    except Exception:
        return {"rocprofv3_median_us": None, "rocprofv3_source": "no profiles/scan_summary.json"}
    sha = csrc_sha256()
    if d.get("csrc_sha256") != sha:
        return {"rocprofv3_median_us": None,
                "rocprofv3_source": f"profiles/scan_summary.json was measured on other kernel sources (csrc sha256 "
                                    f"{str(d.get('csrc_sha256'))[:12]}, this checkout {sha[:12]}): re-run tools/profile_bench.sh"}
    for r in d.get("runs", []):
        if r["bank_rows"] == n_rows and r["queries"] == nq and r["keys"] == keys_mode:
            return {"rocprofv3_median_us": r["kernel_median_ns"] / 1e3, "rocprofv3_frac": r["frac_of_8TBps_median"],
                    "rocprofv3_dispatches": r["dispatches"],
                    "rocprofv3_source": f"profiles/scan_summary.json (rocprofv3 --kernel-trace of tools/scan_bench.py; kernel sources sha256 {sha[:12]} = this checkout)"}
    return {"rocprofv3_median_us": None, "rocprofv3_source": "no run of this configuration in profiles/scan_summary.json"}


def scan_roofline(eng, synth, torch, dev, N, bank):
    """The HBM-bound regime of the path: the keys-only top-k scan (``range_topk_stream``) for a
    handful of queries, END TO END - one call = the stream kernel with the candidate merge as its
    tail (one launch for up to 256 queries).  The product path streams a bf16 copy of the keys
    (N x 512 B per pass), forms approximate similarities and re-ranks the candidates inside its error
    bound with the float32 chain - the result is that of the float32 scan, bit for bit (tests).  One
    pass serves 16 queries, or 32: two groups of 16 share a pass.

      frac = bytes the call STREAMS (passes x N x 512 B for the bf16 copy, x 1 KB for float32 keys)
             / time per call / 8 TB/s - never above 1; ``reference_format_TBps`` (the reference's
             float32 key bytes / time) is given beside it and is not a roofline fraction.
      time per call: 20 calls enqueued back to back between ONE pair of HIP events on the launch
             stream, mean of 3 such measurements.
      frac_of_copy = plain_read_us / us_per_call: ``plain_read_us`` is a kernel that does nothing but
             READ the same bytes in one launch (range_stream_read_timed: 16-byte non-temporal loads,
             32 KB contiguous per workgroup and step), timed the same way - what a launch of that size
             can reach on this chip at all (6.6 TB/s from DRAM, 7 TB/s out of the Infinity Cache).

    Two banks: this run's bank (range_db_large, N = 100 000: its 51 MB bf16 copy stays in the 256 MB
    Infinity Cache between back-to-back calls - ``resident: infinity_cache``) and a keys-only bank
    of N = 1 000 000 generated on the device (512 MB bf16 / 1 GB float32: every call streams it from
    DRAM - ``resident: dram``).  ``product_path``: what ``model.topk`` runs by default."""
    from range_amd import _native
    out = []

    def big_keys(n):
        g = torch.Generator(device=dev).manual_seed(2024)
        k = torch.randn((n, 256), generator=g, device=dev, dtype=torch.float32)
        c = torch.randn((32, 256), generator=g, device=dev, dtype=torch.float32)
        k += 3.0 * c[torch.randint(0, 32, (n,), generator=g, device=dev)]
        return torch.nn.functional.normalize(k, dim=1).contiguous()

    def engine(keys_mode, keys_dev):
        if keys_mode == "f32":
            os.environ["RANGE_TOPKS_KEYS"] = "f32"   # read at range_create
        try:
            e = _native.HipEngine(dev)
        finally:
            os.environ.pop("RANGE_TOPKS_KEYS", None)
        if keys_dev is None:
            e.set_bank(bank.keys, bank.values, bank.xyz, 0)
        else:
            e.set_keys(keys_dev)
        return e

    for n_rows, resident in ((N, "infinity_cache"), (1_000_000, "dram")):
        keys_dev = None if n_rows == N else big_keys(n_rows)
        for keys_mode, sizes in (("bf16", (16, 32, 64)), ("f32", (16,))):
            e = eng if (keys_mode == "bf16" and keys_dev is None) else engine(keys_mode, keys_dev)
            for nq in sizes:
                x = torch.from_numpy(synth.make_queries(nq, seed=11)).to(dev)
                _, e32, _ = eng.encode(x)
                for _ in range(5):
                    e.topk_stream(e32, 16)
                us = sum(e.topk_stream_timed(e32, 16, 20)[2] for _ in range(3)) / 3.0
                groups = (nq + 15) // 16
                per_pass = 1 if groups <= 1 else 2        # range_topk_stream's choice (range_hip.hip)
                passes = (groups + per_pass - 1) // per_pass
                # the same-launch ceiling: a plain kernel that only READS the same bytes, timed the same way
                copy_us = sum(e.stream_read_timed(keys_mode == "f32", passes, 20) for _ in range(3)) / 3.0
                ref_bytes = passes * n_rows * KEY_ROW_BYTES
                streamed = ref_bytes // 2 if keys_mode == "bf16" else ref_bytes
                kname = "topk_stream_bf16_kernel" if keys_mode == "bf16" else "topk_stream_kernel"
                out.append({"kernel": f"{kname}<{per_pass} group(s) of 16 queries per pass> + merge tail (one launch)",
                            "keys": keys_mode, "bank_rows": n_rows, "resident": resident, "queries": nq,
                            "passes": passes, "streamed_bytes": streamed, "us_per_call": us,
                            "us_source": "20 back-to-back calls between one HIP event pair, mean of 3",
                            "streamed_TBps": streamed / (us * 1e-6) / 1e12, "peak_TBps": PEAK_HBM_GBS / 1e3,
                            "frac": streamed / (us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                            "plain_read_us": copy_us, "plain_read_TBps": streamed / (copy_us * 1e-6) / 1e12,
                            "frac_of_copy": copy_us / us,
                            "reference_format_TBps": ref_bytes / (us * 1e-6) / 1e12,
                            "product_path": keys_mode == "bf16",
                            "exact_fallback_queries": e.topk_stream_exact_count(),
                            **scan_rocprof(n_rows, nq, keys_mode)})
            if e is not eng:
                e.close()
        del keys_dev
    # the north star's own sentence: the top-k similarity scan over range_db_large for the 10 000-query
    # batch (``model.topk``).  Beyond 256 queries the scan is GEMM-shaped (range_amd/csrc/topk_gemm.h,
    # round 5): fp16 MFMA products (operands scaled by powers of two) of every (query, key) pair - a sampled pass for per-query thresholds,
    # a full pass that appends the candidates, a float32 re-rank; values and indices are those of the
    # float32 scan bit for bit.  Its bound is the 16-bit MFMA peak, not HBM: the 51 MB of fp16 keys sit in
    # the Infinity Cache.  ``frac`` = the ALGORITHMIC products 2 x 256 x B x N / time / 2.5 PFLOP/s;
    # ``executed_flop`` adds the sampled pass (1 / TG_SAMPLE of a pass).
    nq = 10_000
    x = torch.from_numpy(synth.make_queries(nq, seed=7, lat_max=90.0)).to(dev)
    _, e32, _ = eng.encode(x)
    for _ in range(20):            # (also the pre-heat: 20 x ~0.9 ms)
        eng.topk_stream(e32, 16)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.profile_enable(True)
    t0.record()
    for _ in range(20):
        eng.topk_stream(e32, 16)
    t1.record()
    t1.synchronize()
    us = t0.elapsed_time(t1) / 20 * 1e3
    scan_us, rerank_us = eng.profile_read(_native.PROF_TOPK_STREAM)[0] / 20 * 1e3, eng.profile_read(_native.PROF_TOPK_MERGE)[0] / 20 * 1e3
    eng.profile_enable(False)
    flop = 2.0 * 256 * nq * N
    # pass A visits every tile_stride-th key tile (range_hip.hip: topk_stream_impl - the same integers here)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    n_blocks = (N + 15) // 16
    n_qblocks = (nq + 255) // 256
    n_splits = max(4, min(min(2 * n_cu // n_qblocks, n_blocks // 8), 64))
    tg_sample = max(1, min(16, int(os.environ.get("RANGE_TG_SAMPLE", "5"))))      # (topk_gemm.h: TG_SAMPLE)
    tile_stride = max(1, min(tg_sample, n_blocks // n_splits // 4))
    out.append({"kernel": "topk_gemm_kernel<0> (sampled group maxima) + threshold + topk_gemm_kernel<1> (candidates) + "
                          "topk_gemm_rerank_kernel (float32 re-rank): range_amd/csrc/topk_gemm.h",
                "keys": "fp16", "bank_rows": N, "resident": "infinity_cache", "queries": nq,
                "us_per_call": us, "us_source": "20 calls between one HIP event pair, after 20 untimed ones",
                "us_scan_kernels": scan_us, "us_rerank": rerank_us,
                "bound": "mfma fp16", "algorithmic_flop": flop, "executed_flop": flop * (1.0 + 1.0 / tile_stride),
                "pass_a_tile_stride": tile_stride, "bank_splits": n_splits,
                "achieved_tflops": flop / (us * 1e-6) / 1e12, "peak_tflops": 2500.0,
                "frac": flop / (us * 1e-6) / 1e12 / 2500.0,
                "queries_per_s": nq / (us * 1e-6), "product_path": True,
                # (not measured by this run: BENCH_r04.json's figure for the same batch through the streaming scan)
                "reference_from_BENCH_r04_streaming_scan_us": 3890.0,
                # the 16-bit MFMA rate a register-only loop sustains on random operands (profiles/r06/mfma_f16_peak.log,
                # tools/micro/mfma_f16_peak.hip): what the data sheet's 2.5 PFLOP/s come to on this chip under load
                "sustained_peak_tflops_random_operands": 1890.0, "sustained_peak_source": "profiles/r06/mfma_f16_peak.log",
                "exact_fallback_queries": eng.topk_stream_exact_count()})
    return out


def return_topk_cost(eng, synth, torch, dev, a):
    """``model(coords, return_topk=16)``: the step with the top-k side channel in the SAME call (the
    forward's e-hat serves the scan: range_topk_last) against the plain step, alternating, one HIP event
    pair per leg; and against the two separate calls it replaces (forward + ``model.topk``: a second
    encoder pass).  Never ``value``."""
    from range_amd import _native
    B = a.queries
    x = torch.from_numpy(synth.make_queries(B, seed=7, lat_max=90.0)).to(dev)
    out = torch.empty((B, 1280), dtype=torch.float64, device=dev)

    def plain():
        eng.forward(x, _native.MODEL_RANGE_PLUS, a.beta, out=out)

    def fused():
        eng.forward(x, _native.MODEL_RANGE_PLUS, a.beta, out=out)
        return eng.topk_last(B, 16)

    def separate():
        eng.forward(x, _native.MODEL_RANGE_PLUS, a.beta, out=out)
        _, e32, _ = eng.encode(x)
        return eng.topk_stream(e32, 16)

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n

    for _ in range(8):
        fused()
    n = max(5, a.steps)
    t_plain = t_fused = t_sep = 0.0
    for _ in range(3):
        t_plain += timed(plain, n) / 3
        t_fused += timed(fused, n) / 3
        t_sep += timed(separate, n) / 3
    fv, fi = fused()
    sv, si = separate()
    return {"what": "forward + top-16 of the same 10 000 queries in ONE call (range_topk_last: no second encoder pass)",
            "ms_per_step_plain": t_plain, "ms_per_step_with_topk": t_fused, "added_ms": t_fused - t_plain,
            "added_frac_of_step": (t_fused - t_plain) / t_plain,
            "ms_per_step_two_calls": t_sep, "two_calls_added_ms": t_sep - t_plain,
            "value_with_topk": B / (t_fused * 1e-3),
            "bitwise_equal_to_model_topk": bool(torch.equal(fv, sv) and torch.equal(fi, si))}


def host_contract_rate(eng, synth, torch, dev, beta, B):
    """The reference's own contract: ``model(x)`` returns a host ndarray (B,1280) float64
    (range/range.py:240).  Timed as the caller's loop of synchronous calls, once dropping each
    result before the next call (its memory is recycled, range_amd/_hostpool.py) and once keeping
    every result alive (each call fills fresh pages); ``value`` is the slower.  Never ``value`` of
    the bench line."""
    from range_amd import _native
    xs = [torch.from_numpy(synth.make_queries(B, seed=100 + i)).to(dev) for i in range(4)]
    eng.forward_host(xs[0], _native.MODEL_RANGE_PLUS, beta)
    eng.forward_host(xs[1], _native.MODEL_RANGE_PLUS, beta)
    torch.cuda.synchronize(dev)
    res = {}
    for mode in ("results_dropped", "results_kept"):
        keep = []
        t0 = time.perf_counter()
        n = 0
        for rep in range(2):
            for x in xs:
                r = eng.forward_host(x, _native.MODEL_RANGE_PLUS, beta)
                if mode == "results_kept":
                    keep.append(r)
                n += x.shape[0]
        dt = time.perf_counter() - t0
        res[mode] = {"value": n / dt, "ms_per_batch": dt / (2 * len(xs)) * 1e3}
        del keep
    worst = min(res, key=lambda k: res[k]["value"])
    return {"value": res[worst]["value"], "unit": "geo-embeddings/sec", **res,
            "what": f"{2 * len(xs)} synchronous calls of {B} queries, each returning a host ndarray "
                    "(B,1280) float64; inputs resident in HBM"}


if __name__ == "__main__":
    main()
