#!/usr/bin/env python3
"""PCIe-inclusive throughput of the numpy contract (GPU box only): (a) model(x) per batch with
its synchronous D2H, as the reference's caller does; (b) range_amd.save.EmbeddingPipeline."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from range_amd import load_model
from tools import synth
from range_amd.save import EmbeddingPipeline

tmp = tempfile.mkdtemp()
ck = synth.write_checkpoint(os.path.join(tmp, "e.ckpt"), L=40, hidden=512, seed=1234)
db = synth.write_bank(os.path.join(tmp, "db.npz"), 100_000, 2024)
m = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db, beta=0.5)
nb, B = 12, 10_000
host = [torch.from_numpy(synth.make_queries(B, seed=100 + i)) for i in range(nb)]
dev = [h.cuda() for h in host]
m(dev[0]); list(EmbeddingPipeline(m).run(host[:2]))
torch.cuda.synchronize(); t0 = time.perf_counter()
for x in dev: m(x, return_device=True)
torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
t0 = time.perf_counter()
for x in dev: m(x)
t_sync = time.perf_counter() - t0
t0 = time.perf_counter()
n = sum(o.shape[0] for o in EmbeddingPipeline(m).run(host))
t_pipe = time.perf_counter() - t0
print({"device_resident_q_per_s": nb * B / t_dev, "numpy_contract_sync_q_per_s": nb * B / t_sync,
       "pipelined_host_input_q_per_s": n / t_pipe})
