#!/usr/bin/env python3
"""Forward latency of RANGE+ for small batches on the bench workload (GPU only): device-resident
queries in, device-resident embeddings out, mean of 50 calls after warm-up, and the kernels' own
times (HIP events per kernel).  Up to 32 queries run the one-pass kernel (attend_small.h;
RANGE_SMALL_FORWARD=0 sends them through the two-pass kernels for comparison).
--preheat-ms T: T ms of untimed forwards in front of every batch size (a serving loop under load: the chip
holds its clock only after ~35 ms of continuous work, tools/clock_ramp.py); default 0 = calls arriving at
an idle chip (the protocol of rounds 3-5: 5 warm-up calls, 50 timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from range_amd import _native, sh_table
from tools import synth
from range_amd.bank import prepare_bank

PREHEAT_MS = float(sys.argv[sys.argv.index("--preheat-ms") + 1]) if "--preheat-ms" in sys.argv else 0.0
dev = torch.device("cuda:0")
bank = prepare_bank(*synth.make_bank(100000, 2024))
w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
eng = _native.HipEngine(dev)
eng.set_encoder(40, 512, 2, 256, 0, [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]], sh_table=sh_table.generate_table(40))
eng.set_bank(bank.keys, bank.values, bank.xyz)
for B in (1, 8, 16, 17, 32, 33, 64, 256, 1024, 4096):
    x = torch.from_numpy(synth.make_queries(B, seed=B)).to(dev)
    out = torch.empty((B, 1280), dtype=torch.float64, device=dev)
    for _ in range(5):
        eng.forward(x, _native.MODEL_RANGE_PLUS, 0.5, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if PREHEAT_MS > 0:
        a.record()
        eng.forward(x, _native.MODEL_RANGE_PLUS, 0.5, out=out)
        b.record(); b.synchronize()
        for _ in range(int(PREHEAT_MS / max(a.elapsed_time(b), 0.05)) + 1):
            eng.forward(x, _native.MODEL_RANGE_PLUS, 0.5, out=out)
    a.record()
    for _ in range(50):
        eng.forward(x, _native.MODEL_RANGE_PLUS, 0.5, out=out)
    b.record(); b.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    eng.profile_enable(True)
    for _ in range(20):
        eng.forward(x, _native.MODEL_RANGE_PLUS, 0.5, out=out)
    torch.cuda.synchronize()
    k = {nm: eng.profile_read(i) for i, nm in enumerate(["encoder", "pass1", "pass2_or_onepass"])}
    eng.profile_enable(False)
    ks = ", ".join(f"{nm} {ms / 20 * 1e3:.1f} us/{n // 20}" for nm, (ms, n) in k.items() if n)
    print(f"B={B:5d}: {us:9.1f} us per forward  ({B / us * 1e6:10.0f} geo-embeddings/s); event pairs per forward: {ks}", flush=True)
