#!/usr/bin/env python3
"""Per-workgroup timeline of pass 2 on kept logits (a -DRANGE_EXP_P2_STAMPS build loaded through
RANGE_LIB_PATH with RANGE_ALLOW_EXPERIMENT_BUILD=1; GPU only): prologue / loop / store times per
workgroup, the clock it ran at, and - workgroups ordered per CU - the gap between a workgroup's
end and the start of the next one on the same CU.
Usage: RANGE_P2_STAMPS=/tmp/s.bin python tools/pass2_stamps.py [B] [N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from range_amd import _native
from range_amd.bank import prepare_bank
from tools import synth

path = os.environ["RANGE_P2_STAMPS"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
dev = torch.device("cuda:0")
bank = prepare_bank(*synth.make_bank(N, 2024))
eng = _native.HipEngine(dev)
eng.set_bank(bank.keys, bank.values, bank.xyz)
g = torch.Generator().manual_seed(0)
e32 = torch.nn.functional.normalize(torch.randn(B, 256, generator=g), dim=1).to(dev)
xq = torch.zeros(B, 4)
xq[:, :3] = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
xq = xq.to(dev)
st = eng.scan_stats(e32, xq, 12.0, 40.0, keep_logits=True)
for _ in range(4):
    eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st)
torch.cuda.synchronize()
d = np.fromfile(path, dtype=np.uint64).reshape(-1, 16).astype(np.float64)
c, r = d[:, 0:5], d[:, 5:10] * 0.01          # shader clocks; microseconds (100 MHz)
hw, xcc, nb = d[:, 10].astype(np.int64), d[:, 11].astype(np.int64), d[:, 12]
t0 = r[:, 0].min()
wall = r[:, 4].max() - t0
print(f"B={B} N={N}: {len(d)} workgroups, {nb.mean():.1f} blocks each, kernel wall (first entry to last store done) {wall:.1f} us")
us = lambda i, j: r[:, j] - r[:, i]           # noqa: E731
for nm, i, j in (("prologue (entry -> loop)", 0, 1), ("loop", 1, 2), ("stores issued", 2, 3), ("stores landed", 3, 4), ("whole workgroup", 0, 4)):
    v = us(i, j)
    print(f"  {nm:26s} mean {v.mean():8.2f} us  p5 {np.percentile(v, 5):8.2f}  p50 {np.median(v):8.2f}  p95 {np.percentile(v, 95):8.2f}")
loop_us, loop_clk = us(1, 2), c[:, 2] - c[:, 1]
print(f"  loop: {np.median(loop_us / nb):.3f} us per block, {np.median(loop_clk / nb):.0f} clocks per block, clock {np.median(loop_clk / loop_us) / 1e3:.3f} GHz")
# per-CU timelines: CU identity = (xcc, se, sh?, cu) fields of HW_ID: cu_id bits 8-11, sh 12, se 13-15 (gfx9)
cu = (xcc << 16) | (hw & 0xFF00)
gaps, busy = [], 0.0
first_start, last_end = [], []
for k in np.unique(cu):
    m = np.where(cu == k)[0]
    m = m[np.argsort(r[m, 0])]
    busy += (r[m, 4] - r[m, 0]).sum()
    first_start.append(r[m[0], 0] - t0)
    last_end.append(r[m[-1], 4] - t0)
    gaps += list(r[m[1:], 0] - r[m[:-1], 4])
n_cu = len(np.unique(cu))
print(f"  {n_cu} distinct CUs seen; workgroups per CU {len(d) / n_cu:.2f}; busy share of the wall {busy / (n_cu * wall):.3f}")
if gaps:
    gaps = np.array(gaps)
    print(f"  gap end -> next start on the same CU: mean {gaps.mean():.2f} us  p5 {np.percentile(gaps, 5):.2f}  p50 {np.median(gaps):.2f}  p95 {np.percentile(gaps, 95):.2f}")
print(f"  first start per CU: max {max(first_start):.2f} us after the first; last end per CU: min {min(last_end):.1f}  p50 {np.median(last_end):.1f}  max {max(last_end):.1f} us")
by_round = {}
for k in np.unique(cu):
    m = np.where(cu == k)[0]
    m = m[np.argsort(r[m, 0])]
    for i, w in enumerate(m):
        by_round.setdefault(i, []).append((r[w, 2] - r[w, 1]) / nb[w])
print("  us per block by the workgroup's position on its CU:", {i: round(float(np.median(v)), 3) for i, v in sorted(by_round.items())})
