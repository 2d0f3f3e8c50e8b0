#!/usr/bin/env python3
"""Where pass 2 loses time on a SHORT shard (tuning; GPU only): attend_kept of B queries against N
bank rows for several forced split counts (RANGE_P2_SPLITS, one process per count), and the
instrumented recompute kernel's per-workgroup start / duration stamps (occupancy of the CUs,
per-workgroup cost against its block count).
Usage: python tools/pass2_overhead.py [B] [N] [--diag]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from range_amd import _native
from range_amd.bank import prepare_bank
from tools import synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if len(args) > 0 else 10000
N = int(args[1]) if len(args) > 1 else 12500
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
for _ in range(3):
    eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st)
eng.profile_enable(True)
R = 10
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(R):
    eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st)
b.record()
b.synchronize()
ms_k = eng.profile_read(2)[0] / R
qt, ns = eng.last_geometry()
print(f"B={B} N={N} grid {qt} x {ns} = {qt * ns} workgroups ({qt * ns / 256:.2f} rounds, "
      f"{(N + 15) // 16 / ns:.1f} blocks each): pass-2 kernel {ms_k * 1e3:.1f} us, with reduce {a.elapsed_time(b) / R * 1e3:.1f} us",
      flush=True)
if "--diag" in sys.argv:
    for _ in range(2):
        d = eng.attend_diag(e32, xq, 12.0, 40.0, 0.5, st)
    torch.cuda.synchronize()
    d = d.cpu().numpy().astype(np.float64)[:, 0, :]       # wave 0 of every workgroup
    start, dur, nb = d[:, 9], d[:, 7], d[:, 8]
    t0 = start.min()
    wall = (start + dur).max() - t0
    print(f"  diag (recompute kernel): wall {wall:.0f} ticks, sum of workgroup durations / (256 x wall) = "
          f"{dur.sum() / (256 * wall):.3f}; first-start spread {np.sort(start)[min(255, len(start) - 1)] - t0:.0f}")
    A = np.stack([nb, np.ones_like(nb)], 1)
    c, o = np.linalg.lstsq(A, dur, rcond=None)[0]
    print(f"  workgroup duration ~ {c:.1f} ticks per block + {o:.0f} (blocks per workgroup {nb.min():.0f}..{nb.max():.0f}; "
          f"mean duration {dur.mean():.0f}, p5 {np.percentile(dur, 5):.0f}, p95 {np.percentile(dur, 95):.0f})")
    inloop = d[:, 0] + d[:, 1] + d[:, 2] + d[:, 3] + d[:, 4] + d[:, 5] + d[:, 6]
    print(f"  of which inside the block loop (mean) {inloop.mean():.0f}; outside (prologue + epilogue) {(dur - inloop).mean():.0f}")
    order = np.argsort(start)
    ends = np.sort(start + dur)
    # a workgroup of a later round starts when some earlier one has ended: gap = its start - the
    # (i - 255)-th end
    gaps = [start[order[i]] - ends[i - 256] for i in range(256, len(order))]
    if gaps:
        print(f"  dispatch gap behind the workgroup whose CU it takes: mean {np.mean(gaps):.0f} ticks, p50 {np.median(gaps):.0f}, p95 {np.percentile(gaps, 95):.0f}")
