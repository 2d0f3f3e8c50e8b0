#!/usr/bin/env python3
"""Where the time of the numpy contract goes (GPU box): model(x) returning a fresh host array,
with results kept alive / dropped per call, and the library's own phase times
(RANGE_HOST_TIMING=1), for different host-thread counts and with / without kernel prefaulting."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from range_amd import _native
    from tools import synth
    from range_amd.bank import prepare_bank
    from range_amd.ckpt import EncoderParams
    w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
    bank = prepare_bank(*synth.make_bank(100_000, 2024))
    eng = _native.HipEngine("cuda:0")
    eng.set_encoder(40, 512, 2, 256, 0, [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                    [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])
    eng.set_bank(bank.keys, bank.values, bank.xyz)
    xs = [torch.from_numpy(synth.make_queries(10_000, seed=100 + i)).cuda() for i in range(4)]
    eng.forward_host(xs[0], 1, 0.5); eng.forward_host(xs[1], 1, 0.5)
    for keep_alive in (True, False):
        keep = []
        t0 = time.perf_counter()
        for rep in range(3):
            for x in xs:
                r = eng.forward_host(x, 1, 0.5)
                if keep_alive: keep.append(r)
        dt = (time.perf_counter() - t0) / 12
        print(f"  results {'kept' if keep_alive else 'dropped'}: {dt*1e3:.2f} ms per call = {10_000/dt:,.0f} /s")
    sys.exit(0)
for env in ({}, {"RANGE_HOST_PREFAULT": "0"}, {"RANGE_HOST_THREADS": "8"}, {"RANGE_HOST_THREADS": "32"}):
    print(env or "default")
    e = dict(os.environ, **env)
    subprocess.run([sys.executable, __file__, "child"], env=e)
