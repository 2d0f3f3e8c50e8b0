#!/usr/bin/env python3
"""Timing of the ridge probe (range_amd/evaluate.py) on one MI355X.

    python tools/probe_bench.py [--rows 100000] [--dim 1280] [--classes 16] [--cpu-rows 8000]

Prints one JSON object per task kind with
  * end-to-end seconds of RidgeProbe.fit_score (host arrays in, score out; H2D included) and of
    its stages (HIP events on the stream the kernels run on),
  * the float64 MFMA roofline of the dominant kernel, the per-fold Gram GEMM: algorithmic FLOPs =
    rows * d * (d + 128-tile diagonal overhang excluded) ... i.e. 2 * rows * d*(d+1)/2 for Z^T Z
    plus 2 * rows * d * c for Z^T T, over its measured time, against 78.6 TFLOP/s dense f64,
  * the CPU baseline: the oracle restatement of scikit-learn's path (oracle/probe_oracle.py) on a
    bounded row sample, with the rows/s it reaches.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from range_amd import evaluate as ev  # noqa: E402
from tools import synth  # noqa: E402

F64_MFMA_PEAK = 78.6e12      # MI355X dense FP64 matrix, FLOP/s


def timed(fn, stream):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    out = fn()
    b.record(stream)
    b.synchronize()
    return out, a.elapsed_time(b) * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--val-rows", type=int, default=20000)
    ap.add_argument("--dim", type=int, default=1280)
    ap.add_argument("--classes", type=int, default=16)
    ap.add_argument("--cpu-rows", type=int, default=8000)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev)
    for kind in ("regression", "classification"):
        t = synth.make_probe_task(kind, args.rows, args.val_rows, args.dim, seed=500,
                                  n_classes=args.classes if kind == "classification" else 0)
        probe = ev.RidgeProbe(dev)
        cls = kind == "classification"
        cv = 10 if cls else 3
        probe.fit_score(t["train_embeddings"][:4000], t["train_y"][:4000],
                        t["val_embeddings"][:1000], t["val_y"][:1000], cls)      # warm-up
        torch.cuda.synchronize()
        best = None
        for _ in range(args.repeat):
            t0 = time.perf_counter()
            r = probe.fit_score(t["train_embeddings"], t["train_y"], t["val_embeddings"],
                                t["val_y"], cls)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        stages = {}
        probe.fit_score(t["train_embeddings"], t["train_y"], t["val_embeddings"], t["val_y"], cls,
                        timings=stages)

        # the dominant kernel on its own: Gram statistics of one fold-sized block, device resident
        eng = probe.engine
        d = args.dim
        c = 1 if not cls else (1 if args.classes == 2 else args.classes)
        rows = args.rows // cv
        Z = torch.randn(rows, d, dtype=torch.float64, device=dev)
        T = torch.randn(rows, c, dtype=torch.float64, device=dev)
        G, B = eng.empty((d, d)), eng.empty((d, c))
        zs, ts = eng.empty(d), eng.empty(c)
        eng.gram(Z, T, G, B, zs, ts)
        torch.cuda.synchronize()
        _, g_sec = timed(lambda: [eng.gram(Z, T, G, B, zs, ts) for _ in range(5)], stream)
        g_sec /= 5
        flops = 2.0 * rows * (d * (d + 1) / 2 + d * c)
        # all (fold, alpha) solves
        Gf = torch.stack([G + G.T - torch.diag(torch.diagonal(G))] * cv).contiguous()
        Bf = torch.stack([B] * cv).contiguous()
        zf, tf = torch.stack([zs] * cv).contiguous(), torch.stack([ts] * cv).contiguous()
        Gt, Bt, zt, tt = eng.sum_parts(Gf), eng.sum_parts(Bf), eng.sum_parts(zf), eng.sum_parts(tf)
        ntr = [float(rows * (cv - 1))] * cv
        eng.solve(Gt, Bt, zt, tt, ntr, probe.alphas, Gf, Bf, zf, tf)
        _, s_sec = timed(lambda: eng.solve(Gt, Bt, zt, tt, ntr, probe.alphas, Gf, Bf, zf, tf), stream)
        del Z, T, Gf, Bf

        # CPU baseline: the oracle on a bounded sample of the same task
        from oracle import probe_oracle as po   # checker / baseline leg only
        m = min(args.cpu_rows, args.rows)
        mv = max(1, m // 5)
        t0 = time.perf_counter()
        o = po.probe(t["train_embeddings"][:m], t["train_y"][:m], t["val_embeddings"][:mv],
                     t["val_y"][:mv], kind)
        cpu_sec = time.perf_counter() - t0
        print(json.dumps({
            "task": kind, "train_rows": args.rows, "val_rows": args.val_rows, "dim": d,
            "targets": c, "folds": cv, "alphas": list(probe.alphas),
            "score": r["score"], "alpha": r["alpha"],
            "fit_score_sec": best, "train_rows_per_sec": args.rows / best,
            "stage_sec": {k: round(v, 5) for k, v in stages.items()},
            "gram_kernel": {"rows": rows, "sec": g_sec, "flop": flops,
                            "achieved_tflops": flops / g_sec / 1e12, "peak_tflops": 78.6,
                            "frac": flops / g_sec / F64_MFMA_PEAK, "bound": "mfma(f64)"},
            "solve_all_systems_sec": s_sec, "systems": cv * len(probe.alphas),
            "cpu_baseline": {"kind": "port", "rows": m, "sec": cpu_sec,
                             "train_rows_per_sec": m / cpu_sec,
                             "threads": torch.get_num_threads(), "score": o["score"]},
        }), flush=True)


if __name__ == "__main__":
    main()
