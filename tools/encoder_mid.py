#!/usr/bin/env python3
"""The encoder for 513 .. 2 048 queries (and a 10 000-query batch's last partial round) as ONE
persistent launch (round 5) against the three launches before (RANGE_ENC_FUSED_MID=0): HIP-event time
of the encoder launches, steady state, and the outputs bit for bit.  GPU only.
Usage: python tools/encoder_mid.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from range_amd import _native, sh_table
from tools import synth

dev = torch.device("cuda:0")
w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
Ws = [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]]
bs = [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]]
TABLE = sh_table.generate_table(40)
sizes = [int(v) for v in sys.argv[1:] if v.isdigit()] or [513, 625, 1250, 1808, 2048, 10000]
engs = {}
for mid in ("1", "0"):
    os.environ["RANGE_ENC_FUSED_MID"] = mid
    engs[mid] = _native.HipEngine(dev)
    engs[mid].set_encoder(40, 512, 2, 256, 0, Ws, bs, sh_table=TABLE)
for B in sizes:
    x = torch.from_numpy(synth.make_queries(B, seed=B, lat_max=90.0)).to(dev)
    out, us = {}, {}
    for mid, eng in engs.items():
        for _ in range(max(30, int(150e-3 / (B * 6e-8 + 1e-4)))):        # pre-heat: ~150 ms
            out[mid] = eng.encode(x)
        torch.cuda.synchronize()
        eng.profile_enable(True)
        for _ in range(50):
            eng.encode(x)
        torch.cuda.synchronize()
        us[mid] = eng.profile_read(_native.PROF_ENCODER)[0] / 50 * 1e3
        eng.profile_enable(False)
    same = all(torch.equal(a, b) for a, b in zip(out["1"], out["0"]))
    print(f"B={B:6d}: one launch {us['1']:7.1f} us | three launches {us['0']:7.1f} us | outputs bit-identical: {same}", flush=True)
