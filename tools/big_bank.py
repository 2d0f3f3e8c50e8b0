#!/usr/bin/env python3
"""Robustness at ten times the benchmark's bank (GPU only): N = 1 000 000 rows (values 4.1 GB,
kept logits 4 B per pair), a few thousand queries through load_model-level calls; every row through
the planted-constant-column property, a sample against the float64 oracle, both arithmetic modes
of pass 2, top-16 against the oracle.  Usage: python tools/big_bank.py [N [B]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import range_oracle as O
from range_amd import _native
from tools import synth

args = [int(v) for v in sys.argv[1:] if v.isdigit()]
N = args[0] if args else 1_000_000
B = args[1] if len(args) > 1 else 4096
t0 = time.time()
locs, vals, keys = synth.make_bank(N, 2024)
vals = vals.copy()
vals[:, 0] = 1.0
vals[:, 1] = -2.5
bank = O.prep_bank(locs, vals, keys)
print(f"bank of {N} rows generated in {time.time() - t0:.1f} s", flush=True)
w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
eng = _native.HipEngine("cuda:0")
eng.set_encoder(40, 512, 2, 256, 0, [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])
eng.set_bank(bank.keys, bank.values, bank.xyz)
q = synth.make_queries(B, seed=7)
x = torch.from_numpy(q).cuda()
idx = np.linspace(0, B - 1, 16).astype(np.int64)
ref64 = None
for mode in ("exact", "bf16x3"):
    eng.set_pv_mode(mode)
    out = eng.forward(x, _native.MODEL_RANGE_PLUS, 0.5)
    torch.cuda.synchronize()
    t0 = time.time()
    out = eng.forward(x, _native.MODEL_RANGE_PLUS, 0.5)
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert bool(torch.isfinite(out).all())
    c0 = float((out[:, 0] - 1.0).abs().max()); c1 = float((out[:, 1] + 2.5).abs().max())
    got = out[torch.from_numpy(idx).cuda()].cpu().numpy()
    if ref64 is None:
        ref64 = O.retrieve64(got[:, 1024:], q[idx], bank, "RANGE+", 0.5)
    err = float(np.abs(got[:, :1024] - ref64).max())
    print(f"{mode:7s}: {dt * 1e3:8.1f} ms per forward of {B} queries ({B / dt:9.0f}/s), kept logits for {eng.kept_queries()} queries, "
          f"grid {eng.last_geometry()}, planted columns off by {c0:.2e} / {c1:.2e}, sample vs f64 oracle {err:.2e}", flush=True)
    assert c0 < 5e-5 and c1 < 1e-4 and err < 2e-5
eng.set_pv_mode("exact")
e64, e32, xq = eng.encode(x[torch.from_numpy(idx).cuda()])
tv, ti = eng.topk_stream(e32, 16)
s64, _ = O.logits64(e64.cpu().numpy(), q[idx], bank)
rv, ri = O.topk64(s64, 16)
print("top-16 by the streaming kernel: indices equal", bool(np.array_equal(ti.cpu().numpy(), ri)),
      "max value diff %.2e" % float(np.abs(tv.cpu().numpy() - rv).max()))
