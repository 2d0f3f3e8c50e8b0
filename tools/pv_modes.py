#!/usr/bin/env python3
"""Pass 2 in both arithmetic modes on the bench workload (GPU only): HIP-event time of the kernel,
executed rate, and the error of each against the float64 oracle on a sample of the queries.
Usage: python tools/pv_modes.py [B [N]] [--json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import range_oracle as O
from range_amd import _native, sh_table
from tools import synth

args = [v for v in sys.argv[1:] if v.isdigit()]
B = int(args[0]) if args else 10000
N = int(args[1]) if len(args) > 1 else 100000
dev = torch.device("cuda:0")
bank = O.prep_bank(*synth.make_bank(N, 2024))
w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
eng = _native.HipEngine(dev)
eng.set_encoder(40, 512, 2, 256, 0, [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]], sh_table=sh_table.generate_table(40))
eng.set_bank(bank.keys, bank.values, bank.xyz)
q = synth.make_queries(B, seed=7, lat_max=90.0)
x = torch.from_numpy(q).to(dev)
e64, e32, xq = eng.encode(x)
st = eng.scan_stats(e32, xq, 12.0, 40.0, keep_logits=True)
sample = np.linspace(0, B - 1, 64).astype(np.int64)
ref64 = O.retrieve64(e64.cpu().numpy()[sample], q[sample], bank, "RANGE+", 0.5)
out = {}
for mode in ("exact", "bf16x3"):
    eng.set_pv_mode(mode)
    for _ in range(2):
        part = eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    for _ in range(5):
        part = eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st)
    torch.cuda.synchronize()
    ms = eng.profile_read(_native.PROF_ATTEND)[0] / 5
    eng.profile_enable(False)
    got = part.cpu().numpy()[sample]
    err = np.abs(got - ref64)
    out[mode] = {"pass2_ms": round(ms, 3), "pairs_per_s": round(B * N / ms * 1e3),
                 "max_abs_err_vs_f64": float(err.max()), "mean_abs_err_vs_f64": float(err.mean()),
                 "grid": list(eng.last_geometry())}
    if mode == "bf16x3":
        out[mode]["max_abs_diff_vs_exact"] = float(np.abs(got - exact).max())
        out[mode]["bf16_tflops"] = round(B * N * 1024 * 2 * 6 / ms / 1e9, 1)
    else:
        exact = got
        out[mode]["f32_tflops"] = round(B * N * 1024 * 2 / ms / 1e9, 1)
if "--json" in sys.argv:
    print(json.dumps({"queries": B, "bank_rows": N, **out}))
else:
    for k, v in out.items():
        print(k, v)
