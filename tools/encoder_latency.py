#!/usr/bin/env python3
"""Encoder (kernel A) time against the batch size, reference-faithful table mode and exact
recurrence, small-batch split on / off (GPU only).  HIP-event time of the encoder launch(es) from
the library's profile slots, mean of 20 calls.
Usage: python tools/encoder_latency.py [B ...]"""
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
sizes = [int(v) for v in sys.argv[1:] if v.isdigit()] or [16, 64, 256, 625, 1250, 2500, 4096, 5000, 10000]
for split in ("1", "0"):
    os.environ["RANGE_ENC_SPLIT"] = split
    for mode, table in (("reference table", TABLE), ("recurrence", None)):
        eng = _native.HipEngine(dev)
        eng.set_encoder(40, 512, 2, 256, 0, Ws, bs, sh_table=table)
        row = []
        for B in sizes:
            x = torch.from_numpy(synth.make_queries(B, seed=B, lat_max=90.0)).to(dev)
            for _ in range(3):
                eng.encode(x)
            torch.cuda.synchronize()
            eng.profile_enable(True)
            for _ in range(20):
                eng.encode(x)
            torch.cuda.synchronize()
            ms = eng.profile_read(_native.PROF_ENCODER)[0] / 20
            eng.profile_enable(False)
            row.append(f"{B}: {ms * 1e3:6.1f}")
        print(f"split={split} {mode:16s} us | " + " | ".join(row), flush=True)
        del eng
