#!/usr/bin/env python3
"""Largest |e-hat(HIP) - e-hat(reference)| per latitude band on the reference's pole-to-pole fixture
(tests/golden/latitude_L40_H512_n2.npz): sh_eval='reference' (the default) and 'exact'.  The numbers
INTEGRATION.md quotes and tests/test_gpu_parity.py::test_encoder_reference_mode_over_all_latitudes asserts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from range_amd import _native, sh_table
from tools import synth

z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "latitude_L40_H512_n2.npz"))
q, L, H = z["lonlat"], int(z["L"]), int(z["hidden"])
w = synth.make_encoder_weights(L, H, 256, int(z["num_hidden_layers"]), int(z["seed"]))
ws = [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]]
bs = [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]]
al = np.abs(q[:, 1])
for name, table in (("reference", sh_table.generate_table(L)), ("exact", None)):
    eng = _native.HipEngine("cuda:0")
    eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, ws, bs, sh_table=table)
    e = eng.encode(torch.from_numpy(q).cuda())[0].cpu().numpy()
    d = np.abs(e - z["embedding"]).max(axis=1)
    print(name, " ".join(f"|lat| {lo}-{hi}: {d[(al >= lo) & (al < hi)].max():.2e}" for lo, hi in
                         ((0, 30), (30, 45), (45, 60), (60, 75), (75, 90.1))),
          f"| reference's own batch-vs-single spread 60-75: {z['self_spread'][(al >= 60) & (al < 75)].max():.1e}")
