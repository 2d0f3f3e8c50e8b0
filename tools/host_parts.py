"""Pass-2 time of range_forward_host's shrinking parts against the one-launch forward (HIP events)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from range_amd import _native
from tools import synth
from range_amd.bank import prepare_bank
from range_amd.range import sh_table_for
dev = torch.device("cuda:0")
w, enc = bench.encoder_params(40, 512)
bank = prepare_bank(*synth.make_bank(100_000, 2024))
eng = _native.HipEngine(dev)
eng.set_encoder(40, 512, 2, 256, _native.SH_ANALYTIC, enc.weights, enc.biases, sh_table=sh_table_for(enc))
eng.set_bank(bank.keys, bank.values, bank.xyz, 0)
x = torch.from_numpy(synth.make_queries(10_000, seed=7, lat_max=90.0)).to(dev)
for name, fn in (("forward (device result)", lambda: eng.forward(x, 1, 0.5)), ("forward_host", lambda: eng.forward_host(x, 1, 0.5))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    per = []
    for _ in range(30):                      # per-call times: the box's clocks wander by +-0.2 ms
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); per.append(time.perf_counter() - t0)
    per.sort()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    a, n = eng.profile_read(_native.PROF_ATTEND); s, _ = eng.profile_read(_native.PROF_SCAN_STATS); e, _ = eng.profile_read(_native.PROF_ENCODER)
    eng.profile_enable(False)
    print(f"{name}: {dt*1e3:.2f} ms per call (of 30 single calls: median {per[15]*1e3:.2f}, fastest quarter {per[7]*1e3:.2f}); pass 2 {a/10:.2f} ms in {n//10} launches, pass 1 {s/10:.2f}, encoder {e/10:.2f}")
