#!/usr/bin/env python3
"""The two evaluations of the 'analytic' spherical harmonics on the GPU (GPU box): per latitude
band, the distance of the L2-normalised embedding from the REFERENCE's own numbers
(tests/golden/latitude_L40_H512_n2.npz) for sh_eval='exact' (stable recurrence) and
sh_eval='reference' (the generated polynomials), next to the reference's own spread; and the
encoder kernel's time for 10 000 queries in both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from range_amd import _native, sh_table
from tools import synth

z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "latitude_L40_H512_n2.npz"))
q, L, H = z["lonlat"], int(z["L"]), int(z["hidden"])
w = synth.make_encoder_weights(L, H, 256, 2, int(z["seed"]))
ws = [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]]
bs = [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]]
table = sh_table.generate_table(L)
al = np.abs(q[:, 1])
x10k = torch.from_numpy(synth.make_queries(10_000, seed=7, lat_max=90.0)).cuda()
res = {}
for name, t in (("exact", None), ("reference", table)):
    eng = _native.HipEngine("cuda:0")
    eng.set_encoder(L, H, 2, 256, _native.SH_ANALYTIC, ws, bs, sh_table=t)
    e = eng.encode(torch.from_numpy(q).cuda())[0].cpu().numpy()
    res[name] = np.abs(e - z["embedding"]).max(axis=1)
    for _ in range(3): eng.encode(x10k)
    eng.profile_enable(True)
    for _ in range(10): eng.encode(x10k)
    torch.cuda.synchronize()
    ms, n = eng.profile_read(_native.PROF_ENCODER)
    print(f"sh_eval={name:9s}: encoder {ms / n:.3f} ms per 10 000 queries")
print("max |embedding - reference's| by |lat| band:   exact      reference   (reference's own spread)")
for lo, hi in ((0, 30), (30, 45), (45, 60), (60, 75), (75, 90.1)):
    m = (al >= lo) & (al < hi)
    print(f"  {lo:4.0f}-{hi:4.0f}  n={int(m.sum()):3d}   {res['exact'][m].max():.2e}   {res['reference'][m].max():.2e}   ({z['self_spread'][m].max():.2e})")
