#!/usr/bin/env python3
"""Randomised cross-check (GPU): for random bank sizes / batch sizes / temperatures, pass 2 on the
kept logits must equal the recomputing pass 2 bit for bit, the top-k from the kept logits must
equal the in-scan top-k (RANGE_KEEP_LOGITS=0 context), and forward() must agree between the two
kinds of context.  Usage: python tools/fuzz_kept.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from range_amd import _native
from tools import synth
from range_amd.bank import prepare_bank

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
w = synth.make_encoder_weights(10, 64, 256, 2, 5)
ws = [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]]
bs = [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]]


def engine(keep, bank):
    os.environ["RANGE_KEEP_LOGITS"] = "1" if keep else "0"
    e = _native.HipEngine("cuda:0")
    e.set_encoder(10, 64, 2, 256, 0, ws, bs)
    e.set_bank(bank.keys, bank.values, bank.xyz)
    return e


bad = 0
for case in range(cases):
    N = int(rng.choice([1, 3, 15, 16, 17, 31, 100, 257, 1000, 1024, 4099, 12345, 30000, 60001]))
    B = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 100, 255, 256, 257, 700, 1500, 3000]))
    bank = prepare_bank(*synth.make_bank(N, int(rng.integers(1 << 30))))
    ek, e0 = engine(True, bank), engine(False, bank)
    x = torch.from_numpy(synth.make_queries(B, seed=int(rng.integers(1 << 30)), lat_max=80.0)).cuda()
    _, e32, xq = ek.encode(x)
    tau_geo = float(rng.choice([0.0, 40.0]))
    tau = float(rng.choice([12.0, 15.0, 3.0]))
    beta = float(rng.choice([0.0, 0.3, 1.0])) if tau_geo > 0 else 1.0
    st = ek.scan_stats(e32, xq, tau, tau_geo, keep_logits=True)
    ok = ek.kept_queries() == B
    a = ek.attend(e32, xq, tau, tau_geo, beta, st)
    ok &= torch.equal(a, ek.attend_kept(0, xq, tau, tau_geo, beta, st))
    if B > 64:
        f = 64 * int(rng.integers(1, (B + 63) // 64))
        n = int(rng.integers(1, B - f + 1))
        ok &= torch.equal(ek.attend(e32[f:f + n], xq[f:f + n], tau, tau_geo, beta, st[f:f + n]),
                          ek.attend_kept(f, xq[f:f + n], tau, tau_geo, beta, st[f:f + n]))
    k = int(rng.integers(1, 17))
    _, tv, ti = ek.scan_stats(e32, xq, tau, tau_geo, topk=k)
    _, tv0, ti0 = e0.scan_stats(e32, xq, tau, tau_geo, topk=k)
    ok &= torch.equal(ti, ti0) and torch.equal(tv, tv0)
    # the side channel's own kernels: the streaming scan (up to 256 queries) / the GEMM-shaped path
    # (topk_gemm.h, beyond) must return the same values and rows
    sv, si = ek.topk_stream(e32, k)
    ok &= torch.equal(si, ti) and torch.equal(sv, tv)
    m = _native.MODEL_RANGE_PLUS if tau_geo > 0 else _native.MODEL_RANGE
    ok &= torch.equal(ek.forward(x, m, beta), e0.forward(x, m, beta))
    print(f"case {case:3d}: N={N:6d} B={B:5d} tau={tau} geo={tau_geo} beta={beta} k={k:2d} {'ok' if ok else 'MISMATCH'}",
          flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
