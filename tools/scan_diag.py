#!/usr/bin/env python3
"""Where a range_topk_stream call's time goes: scan kernel / merge (separate launches with
RANGE_TOPKS_FUSED=0, HIP-event pairs per kernel), for queries unrelated to the keys (the bench's)
and queries that sit inside a crowd of similar keys.  RANGE_TOPKS_DIAG=1 prints candidate counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from range_amd import _native
from tools.scan_bench import make_keys

dev = torch.device("cuda:0")
os.environ["RANGE_TOPKS_DIAG"] = "1"
for n in (100_000, 1_000_000):
    keys = make_keys(n, dev)
    for fused in ("1", "0"):
        os.environ["RANGE_TOPKS_FUSED"] = fused
        eng = _native.HipEngine(dev)
        eng.set_keys(keys)
        for nq in (16, 64):
            g = torch.Generator(device=dev).manual_seed(nq)
            crowd = torch.nn.functional.normalize(keys[torch.randint(0, n, (nq,), generator=g, device=dev)]
                                                  + 0.5 * torch.randn((nq, 256), generator=g, device=dev), dim=1).contiguous()
            free = torch.nn.functional.normalize(torch.randn((nq, 256), generator=g, device=dev), dim=1).contiguous()
            for name, e32 in (("unrelated", free), ("crowded", crowd)):
                for _ in range(3): eng.topk_stream(e32, 16)
                eng.profile_enable(True)
                for _ in range(20): eng.topk_stream(e32, 16)
                torch.cuda.synchronize()
                ms, cnt = eng.profile_read(_native.PROF_TOPK_STREAM)
                mms, mcnt = eng.profile_read(_native.PROF_TOPK_MERGE)
                eng.profile_enable(False)
                us = sum(eng.topk_stream_timed(e32, 16, 20)[2] for _ in range(3)) / 3
                print(f"N={n} fused={fused} q={nq} {name}: call {us:.1f} us (20 back to back); event pair per launch: "
                      f"scan {ms / cnt * 1e3:.1f} us, merge {mms / max(mcnt, 1) * 1e3:.1f} us", flush=True)
                eng.topk_stream_exact_count()
        eng.close()
