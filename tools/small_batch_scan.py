#!/usr/bin/env python3
"""HBM-bound regime of the keys scan (north star: "top-k similarity scan ... HBM GB/s against the
roofline"): tiny query batches against range_db_large.  GPU only.  RANGE_TOPKS_KEYS=f32 streams the
float32 keys instead of the default bf16 prefilter."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from range_amd import _native
from tools import synth
from range_amd.bank import prepare_bank
N = 100000
bank = prepare_bank(*synth.make_bank(N, 2024))
eng = _native.HipEngine("cuda:0")
eng.set_bank(bank.keys, bank.values, bank.xyz)
g = torch.Generator().manual_seed(0)
stream_only = "--stream-only" in sys.argv
for B, topk in (((16, 16), (32, 16), (64, 16)) if stream_only else
                ((16, 0), (16, 16), (64, 0), (64, 16), (256, 16), (1024, 16))):
    e32 = torch.nn.functional.normalize(torch.randn(B, 256, generator=g), dim=1).cuda()
    xq = torch.zeros(B, 4); xq[:, :3] = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1); xq = xq.cuda()
    byt = N * 1024
    if not stream_only:
        for _ in range(5): eng.scan_stats(e32, xq, 12.0, 40.0, topk=topk)
        eng.profile_enable(True)
        for _ in range(50): eng.scan_stats(e32, xq, 12.0, 40.0, topk=topk)
        torch.cuda.synchronize()
        ms, n = eng.profile_read(1)
        us = ms / n * 1e3
        print(f"B={B:5d} topk={topk:2d}: scan kernel {us:8.1f} us  keys stream {byt/us/1e6:6.2f} TB/s = {100*byt/us/1e6/8.0:5.1f} % of 8 TB/s")
        eng.profile_enable(False)
    if topk and B <= 64:
        for _ in range(5): eng.topk_stream(e32, topk)
        eng.profile_enable(True)
        for _ in range(50): eng.topk_stream(e32, topk)
        torch.cuda.synchronize()
        ms, n = eng.profile_read(3)
        us = ms / n * 1e3
        mms, mn = eng.profile_read(4)
        groups = (B + 15) // 16
        G = int(os.environ.get('RANGE_TOPKS_GROUPS', '0')) or (1 if groups <= 1 else 2)
        passes = (groups + G - 1) // G
        t0 = time.perf_counter()
        for _ in range(50): eng.topk_stream(e32, topk)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 50 * 1e6
        bf16 = os.environ.get("RANGE_TOPKS_KEYS", "bf16") != "f32"    # the default: bf16 prefilter + float32 re-rank
        sb = byt // 2 if bf16 else byt
        print(f"        B={B}: whole call {wall:6.1f} us; merge kernel {mms / mn * 1e3:5.1f} us;", end="")
        print(f"        stream kernel ({'bf16' if bf16 else 'f32'} keys): {us:8.1f} us  ({passes} key passes)  "
              f"streams {passes*sb/us/1e6:6.2f} TB/s = {100*passes*sb/us/1e6/8.0:5.1f} % of 8 TB/s"
              f" (float32 key bytes / time: {passes*byt/us/1e6:6.2f} TB/s)")
        eng.profile_enable(False)
