#!/usr/bin/env python3
"""Randomised cross-check (GPU) of the ridge probe against the oracle on random small tasks:
odd feature counts, few rows, many classes, several targets.  Usage: fuzz_probe.py [cases] [seed]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import probe_oracle as po          # the checker
from range_amd import evaluate as ev
from tools import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
probe = ev.RidgeProbe("cuda:0")
bad = 0
for case in range(cases):
    kind = str(rng.choice(["regression", "classification"]))
    d = int(rng.choice([1, 2, 7, 63, 64, 65, 100, 129, 300, 640, 1281]))
    n = int(rng.choice([40, 90, 257, 1000, 2500]))
    if kind == "classification":
        n = max(n, 120)
    nv = int(rng.integers(5, 400))
    c = int(rng.choice([2, 3, 5, 17, 60]))
    kt = int(rng.choice([1, 1, 2, 5]))
    t = synth.make_probe_task(kind, n, nv, d, int(rng.integers(1 << 30)), n_classes=c, n_targets=kt,
                              prior_scale=float(rng.choice([1.0, 2.0])))
    if kind == "classification" and np.unique(t["train_y"]).size < 2:
        continue                                                  # a degenerate draw
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            o = po.probe(t["train_embeddings"], t["train_y"], t["val_embeddings"], t["val_y"], kind)
        except Exception as ex:                                   # e.g. a class with too few members
            print(f"case {case:3d}: {kind} n={n} d={d}: oracle raised {type(ex).__name__}, skipped")
            continue
    r = probe.fit_score(t["train_embeddings"], t["train_y"], t["val_embeddings"], t["val_y"],
                        kind == "classification")
    if kind == "classification":
        ok = r["score"] == o["score"] and np.array_equal(r["cv_scores"], o["cv_scores"])
    else:
        ok = abs(r["score"] - o["score"]) < 1e-8 and np.allclose(r["cv_scores"], o["cv_scores"], rtol=0, atol=1e-8)
    ok = ok and r["alpha"] == o["alpha"]
    print(f"case {case:3d}: {kind:14s} n={n:5d} d={d:5d} c/targets={c if kind[0]=='c' else kt:3d} "
          f"score {r['score']:.6f} vs {o['score']:.6f} alpha {r['alpha']} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
