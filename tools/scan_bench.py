#!/usr/bin/env python3
"""The HBM-bound regime of the path, end to end: ``range_topk_stream`` (scan + merge, one call) on
keys-only banks generated on the device - range_db_large's N = 100 000 (its 51 MB bf16 copy lives in
the 256 MB Infinity Cache once warm) and N = 1 000 000 (512 MB bf16 / 1 GB float32: DRAM-resident).
Prints one JSON line per (N, queries, keys): time per call (20 calls between one event pair, mean of
3), bytes streamed / time against 8 TB/s, indices against a float64 top-k computed with torch.
Usage: python tools/scan_bench.py [--n 100000,1000000] [--q 16,64] [--keys bf16,f32] [--cold]
--cold: flush the caches between calls (one call per event pair, a 1 GiB memset in between)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from range_amd import _native


def make_keys(n, dev, seed=2024):
    g = torch.Generator(device=dev).manual_seed(seed)
    k = torch.randn((n, 256), generator=g, device=dev, dtype=torch.float32)
    c = torch.randn((32, 256), generator=g, device=dev, dtype=torch.float32)
    k += 3.0 * c[torch.randint(0, 32, (n,), generator=g, device=dev)]
    return torch.nn.functional.normalize(k, dim=1).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", default="100000,1000000")
    ap.add_argument("--q", default="16,32,64")
    ap.add_argument("--keys", default="bf16,f32")
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--preheat-ms", type=float, default=0.0,
                    help="untimed calls for this long in front of the timed ones (the chip holds its clock only "
                         "after ~35 ms of continuous work: tools/clock_ramp.py); 0 = the protocol of rounds 3-4")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for n in [int(v) for v in a.n.split(",")]:
        keys = make_keys(n, dev)
        for mode in a.keys.split(","):
            if mode == "f32":
                os.environ["RANGE_TOPKS_KEYS"] = "f32"
            eng = _native.HipEngine(dev)
            os.environ.pop("RANGE_TOPKS_KEYS", None)
            eng.set_keys(keys)
            for nq in [int(v) for v in a.q.split(",")]:
                g = torch.Generator(device=dev).manual_seed(nq)
                # queries near the clusters, so that the top of the list is crowded like a real bank's
                e32 = torch.nn.functional.normalize(
                    keys[torch.randint(0, n, (nq,), generator=g, device=dev)]
                    + 0.5 * torch.randn((nq, 256), generator=g, device=dev), dim=1).contiguous()
                for _ in range(3):
                    tv, ti = eng.topk_stream(e32, 16)
                if a.cold:
                    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
                    us_l = []
                    for _ in range(5):
                        junk.fill_(1.0)
                        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        s0.record()
                        eng.topk_stream(e32, 16)
                        s1.record()
                        s1.synchronize()
                        us_l.append(s0.elapsed_time(s1) * 1e3)
                    us = sum(us_l) / len(us_l)
                    del junk
                else:
                    if a.preheat_ms > 0:
                        one = eng.topk_stream_timed(e32, 16, 20)[2]
                        for _ in range(int(a.preheat_ms * 1e3 / max(one * 20, 1.0)) + 1):
                            eng.topk_stream_timed(e32, 16, 20)
                    us = sum(eng.topk_stream_timed(e32, 16, 20)[2] for _ in range(3)) / 3
                s64 = e32.double() @ keys.double().T
                rv, ri = torch.topk(s64, 16, dim=1)
                same = int((ri == ti).all(dim=1).sum())
                groups = (nq + 15) // 16
                per_pass = 1 if groups <= 1 else 2
                passes = (groups + per_pass - 1) // per_pass
                row = 512 if mode == "bf16" else 1024
                streamed = passes * n * row
                if a.preheat_ms > 0:
                    one = eng.stream_read_timed(mode == "f32", passes, 20)
                    for _ in range(int(a.preheat_ms * 1e3 / max(one * 20, 1.0)) + 1):
                        eng.stream_read_timed(mode == "f32", passes, 20)
                copy_us = sum(eng.stream_read_timed(mode == "f32", passes, 20) for _ in range(3)) / 3
                print(json.dumps({"N": n, "queries": nq, "keys": mode, "passes": passes, "us_per_call": round(us, 2),
                                  "plain_read_us": round(copy_us, 2), "frac_of_copy": round(copy_us / us, 4),
                                  "streamed_MB": streamed / 1e6, "streamed_TBps": round(streamed / us / 1e6, 3),
                                  "frac_streamed": round(streamed / us / 1e6 / 8.0, 4),
                                  "cold": a.cold, "preheat_ms": a.preheat_ms, "queries_with_all_16_indices_equal_f64": same,
                                  "max_val_diff": float((rv.float() - tv).abs().max()),
                                  "exact_fallbacks": eng.topk_stream_exact_count()}), flush=True)
                del s64
            eng.close()
        del keys


if __name__ == "__main__":
    main()
