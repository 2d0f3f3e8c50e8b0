#!/usr/bin/env python3
"""Kernel-tuning helper (GPU only): HIP-event times of the three kernels on the bench workload,
no result checks (usable with deliberately broken experiment builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from range_amd import _native
from tools import synth
from range_amd.bank import prepare_bank

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
bank = prepare_bank(*synth.make_bank(N, 2024))
w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
eng = _native.HipEngine(dev)
eng.set_encoder(40, 512, 2, 256, 0, [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])
eng.set_bank(bank.keys, bank.values, bank.xyz)
x = torch.from_numpy(synth.make_queries(B, seed=7)).to(dev)
e64, e32, xq = eng.encode(x)
st = eng.scan_stats(e32, xq, 12.0, 40.0)
st_ok = torch.zeros_like(st); st_ok[:, 0] = 17.0; st_ok[:, 1] = 100.0; st_ok[:, 2] = 57.0; st_ok[:, 3] = 100.0
names = ["encoder", "scan_stats", "attend"]
for kept in (False, True):        # pass 2 recomputing the logits / on the logits pass 1 kept
    def one():
        eng.encode(x)
        eng.scan_stats(e32, xq, 12.0, 40.0, keep_logits=kept)
        if kept:
            eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st_ok)
        else:
            eng.attend(e32, xq, 12.0, 40.0, 0.5, st_ok)
    for _ in range(3):
        one()
    eng.profile_enable(True)
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    print("kept logits" if kept else "recompute  ",
          {n: round(eng.profile_read(i)[0] / steps, 4) for i, n in enumerate(names)})
    eng.profile_enable(False)
