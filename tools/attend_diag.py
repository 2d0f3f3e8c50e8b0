#!/usr/bin/env python3
"""Kernel-tuning diagnostic: where do the attend kernel's wave cycles go?  (GPU only.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from range_amd import _native
from tools import synth
from range_amd.bank import prepare_bank

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda:0")
bank = prepare_bank(*synth.make_bank(N, 2024))
eng = _native.HipEngine(dev)
eng.set_bank(bank.keys, bank.values, bank.xyz)
g = torch.Generator().manual_seed(0)
e32 = torch.nn.functional.normalize(torch.randn(B, 256, generator=g), dim=1).to(dev)
xq = torch.zeros(B, 4); xq[:, :3] = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
xq = xq.to(dev)
st = eng.scan_stats(e32, xq, 12.0, 40.0)
for _ in range(2):
    d = eng.attend_diag(e32, xq, 12.0, 40.0, 0.5, st)
torch.cuda.synchronize()
d = d.cpu().numpy().astype(np.float64)
nb = d[:, :, 8]
names = ["vm0", "bar0", "pv0", "vm1", "bar1", "qk", "pv1", "total"]
print("workgroups", d.shape[0], "blocks/wg", nb.mean())
per_block = d[:, :, :8] / nb[:, :, None]
for i, n in enumerate(names):
    print(f"{n:6s} mean {per_block[:, :, i].mean():9.1f}  p50 {np.median(per_block[:, :, i]):9.1f}  p95 {np.percentile(per_block[:, :, i], 95):9.1f} cycles/block")
start = d[:, :, 9]; tot = d[:, :, 7]
t0 = start.min()
print("kernel span (cycles, 100MHz memtime?):", (start + tot).max() - t0, " first-start spread:", start.max() - t0)
xcc = d[:, 0, 10].astype(int)
print("xcc histogram by blockIdx%8:", [np.bincount(xcc[i::8], minlength=8).tolist() for i in range(8)][:3])
