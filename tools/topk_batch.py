#!/usr/bin/env python3
"""Batch-scale top-k (GPU only): model.topk for B queries on a synthetic bank - time per call, and
the indices / values against the streaming scan (RANGE_TOPK_GEMM=0 in a second engine) and, on a
sample, against the float64 oracle.  Usage: python tools/topk_batch.py [B] [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from range_amd import _native, sh_table
from range_amd.bank import prepare_bank
from tools import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda:0")
w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
locs, vals, keys = synth.make_bank(N, 2024)
bank = prepare_bank(locs, vals, keys)


def engine():
    e = _native.HipEngine(dev)
    e.set_encoder(40, 512, 2, 256, 0, [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                  [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]], sh_table=sh_table.generate_table(40))
    e.set_bank(bank.keys, bank.values, bank.xyz)
    return e


eng = engine()
os.environ["RANGE_TOPK_GEMM"] = "0"
old = engine()
os.environ.pop("RANGE_TOPK_GEMM")
q = synth.make_queries(B, seed=7, lat_max=90.0)
x = torch.from_numpy(q).to(dev)
_, e32, _ = eng.encode(x)
res = {}
for name, e in (("gemm", eng), ("stream", old)):
    for _ in range(3):
        tv, ti = e.topk_stream(e32, 16)
    e.profile_enable(True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    R = 10
    for _ in range(R):
        tv, ti = e.topk_stream(e32, 16)
    b.record()
    b.synchronize()
    res[name] = (tv.cpu().numpy(), ti.cpu().numpy())
    print(f"{name:7s}: {a.elapsed_time(b) / R * 1e3:8.1f} us per call of {B} queries (N={N});  scan kernels "
          f"{e.profile_read(3)[0] / R * 1e3:.1f} us, merge / re-rank {e.profile_read(4)[0] / R * 1e3:.1f} us;  "
          f"brute-force queries so far {e.topk_stream_exact_count()}", flush=True)
    e.profile_enable(False)
same_i = np.array_equal(res["gemm"][1], res["stream"][1])
same_v = np.array_equal(res["gemm"][0], res["stream"][0])
print("gemm == stream: indices", same_i, "values (bitwise)", same_v)
if not same_i:
    bad = np.flatnonzero((res["gemm"][1] != res["stream"][1]).any(axis=1))
    print("  differing queries:", bad[:10], "of", bad.size)
    qq = bad[0]
    print(res["gemm"][1][qq], res["gemm"][0][qq]); print(res["stream"][1][qq], res["stream"][0][qq])
from oracle import range_oracle as O    # checker only
idx = np.linspace(0, B - 1, 256, dtype=np.int64)
obank = O.prep_bank(locs, vals, keys)
s64, _ = O.logits64(e32[torch.from_numpy(idx).to(dev)].cpu().numpy().astype(np.float64), q[idx], obank)
rv, ri = O.topk64(s64, 16)
mism = int((res["gemm"][1][idx] != ri).any(axis=1).sum())
print(f"vs float64 oracle on 256 queries: {mism} queries differ (ties within 4 ulp of float32 allowed), max |value diff| "
      f"{np.abs(res['gemm'][0][idx] - rv).max():.2e}")
