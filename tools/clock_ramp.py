#!/usr/bin/env python3
"""Does a kernel's time depend on how long the GPU has been busy?  (tuning; GPU only.)
Pass 2 on kept logits, B queries x N rows, launched back to back for `iters` iterations with an
event pair around every launch: prints the per-launch time at several points of the run.
Usage: python tools/clock_ramp.py [B] [N] [iters] [idle_ms]   (idle_ms: host sleep between launches)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from range_amd import _native
from range_amd.bank import prepare_bank
from tools import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 400
idle_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
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
torch.cuda.synchronize()
time.sleep(0.5)
eng.profile_enable(True)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
t0 = time.time()
for a, b in ev:
    a.record()
    eng.attend_kept(0, xq, 12.0, 40.0, 0.5, st)
    b.record()
    if idle_ms:
        b.synchronize()
        t_end = time.perf_counter() + idle_ms * 1e-3          # (a spin, not a sleep: sub-millisecond gaps)
        while time.perf_counter() < t_end:
            pass
torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in ev])
cum = np.cumsum(ms)
print(f"B={B} N={N} idle {idle_ms} ms: launch+reduce time (ms) at iteration (cumulative busy ms):")
for i in sorted({0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, iters - 1}):
    if i < iters:
        print(f"  it {i:5d} ({cum[i]:8.1f} ms): {ms[i]:.4f}   mean of last 8 up to here {ms[max(0, i - 7):i + 1].mean():.4f}")
print(f"  kernel-only mean over the run {eng.profile_read(2)[0] / iters:.4f} ms; min launch+reduce {ms.min():.4f}, median {np.median(ms):.4f}")
