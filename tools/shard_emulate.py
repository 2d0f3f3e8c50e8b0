#!/usr/bin/env python3
"""What one rank of an N-GPU row-sharded run computes per step, on ONE GPU and without any
communication: its own 10 000 queries through the encoder, pass 1 and pass 2 for ALL N*10 000
queries against its 100 000/N bank rows (in 4 chunks, as ShardedRange does), finalize of its own
slice.  The time of this against the single-GPU step bounds the scaling efficiency from above.
Usage: python tools/shard_emulate.py [N ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from range_amd import _native, synth
from range_amd.bank import prepare_bank

dev = torch.device("cuda:0")
w = synth.make_encoder_weights(40, 512, 256, 2, 1234)
full = prepare_bank(*synth.make_bank(100000, 2024))
B = 10000
for W in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8]:
    n = 100000 // W
    eng = _native.HipEngine(dev)
    eng.set_encoder(40, 512, 2, 256, 0, [w["layers.0.weight"], w["layers.1.weight"], w["last_layer.weight"]],
                    [w["layers.0.bias"], w["layers.1.bias"], w["last_layer.bias"]])
    eng.set_bank(full.keys[:n], full.values[:n], full.xyz[:n])
    x = torch.from_numpy(synth.make_queries(B, seed=7)).to(dev)
    e64, e32, xq = eng.encode(x)
    e32_all, xq_all = e32.repeat(W, 1).contiguous(), xq.repeat(W, 1).contiguous()
    cuts = [0] + [((B * c) // 4 + 32) // 64 * 64 for c in (1, 2, 3)] + [B]

    def step():
        eng.encode(x)
        st = eng.scan_stats(e32_all, xq_all, 12.0, 40.0, keep_logits=True)
        st = eng.merge_stats(st[None])
        outs = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            first, m = W * lo, W * (hi - lo)
            part = eng.attend_kept(first, xq_all[first:first + m], 12.0, 40.0, 0.5, st[first:first + m])
            outs.append(eng.finalize(part.reshape(W, hi - lo, 1024), e64[lo:hi].contiguous()))
        return outs

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eng.profile_enable(True)
    a.record()
    for _ in range(5):
        step()
    b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 5
    k = {nm: round(eng.profile_read(i)[0] / 5, 3) for i, nm in enumerate(["encoder", "scan_stats", "attend"])}
    qt, ns = eng.last_geometry()
    print(f"N={W}: {ms:7.3f} ms per step and rank  kernels {k}  last pass-2 grid {qt} x {ns}", flush=True)
    del eng
