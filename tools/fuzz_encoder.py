#!/usr/bin/env python3
"""Randomised cross-check (GPU) of the fused encoder against the float64 oracle: random L, hidden
width (4- and 16-wave kernels), depth, SH convention and batch size (1- and 2-tile workgroups,
half-size last rounds).  Usage: fuzz_encoder.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import range_oracle as O           # the checker
from range_amd import _native
from tools import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    L = int(rng.choice([1, 2, 3, 7, 10, 16, 25, 40]))
    H = int(rng.choice([64, 128, 192, 256, 320, 384, 448, 512]))
    layers = int(rng.choice([1, 2, 3]))
    mode = str(rng.choice(["analytic", "closed-form"]))
    B = int(rng.choice([1, 15, 16, 17, 33, 100, 1000, 4096, 4097, 8193, 9000, 12289, 20000]))
    seed = int(rng.integers(1 << 30))
    w = synth.make_encoder_weights(L, H, 256, layers, seed)
    ws = [w[f"layers.{i}.weight"] for i in range(layers)] + [w["last_layer.weight"]]
    bs = [w[f"layers.{i}.bias"] for i in range(layers)] + [w["last_layer.bias"]]
    eng = _native.HipEngine("cuda:0")
    eng.set_encoder(L, H, layers, 256, _native.SH_ANALYTIC if mode == "analytic" else _native.SH_CLOSED_FORM, ws, bs)
    q = synth.make_queries(B, seed=seed % 1000, lat_max=89.0)
    e64, e32, xq = eng.encode(torch.from_numpy(q).cuda())
    raw = eng.encode_raw(torch.from_numpy(q).cuda())
    ref = O.siren_forward(O.sh_features(q, L, mode), w)          # un-normalised SirenNet output
    err_raw = float(np.abs(raw.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
    refn = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    err = float(np.abs(e64.cpu().numpy() - refn).max())
    ok = err < 5e-12 and err_raw < 5e-12 and np.array_equal(e32.cpu().numpy(), e64.cpu().numpy().astype(np.float32))
    print(f"case {case:3d}: L={L:2d} H={H:3d} layers={layers} {mode:11s} B={B:5d} err {err:.1e} raw {err_raw:.1e} "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
