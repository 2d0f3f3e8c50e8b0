#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU at full size: RANGE+ embeddings of 10^6 queries (range_db_large) for
beta in {0, .25, .5, .75, 1}, device-resident in and out (5 x 10^6 x 1280 float64 = 51 GB of
results).  ``sweep()`` runs one pass 1 and two passes 2 per chunk and blends per beta; the
alternative is one forward per beta.  A 32-row sample of every beta is checked against the float64
oracle (checker only)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from range_amd import load_model
from tools import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
betas = (0.0, 0.25, 0.5, 0.75, 1.0)
tmp = os.environ.get("TMPDIR", "/tmp")
ck = synth.write_checkpoint(os.path.join(tmp, "sweep.ckpt"), L=40, hidden=512, seed=1234)
locs, vals, keys = synth.make_bank(synth.BANK_ROWS["range_db_large"], 2024)
db = os.path.join(tmp, "sweep_db.npz")
np.savez(db, locs=locs, image_embeddings=vals, satclip_embeddings=keys)
m = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db, beta=0.5)
q = synth.make_queries(B, seed=7)
x = torch.from_numpy(q).to("cuda:0")
m.sweep(x[:20000], betas, return_device=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
sw = m.sweep(x, betas, return_device=True)
torch.cuda.synchronize()
t_sweep = time.perf_counter() - t0
t0 = time.perf_counter()
for b in betas:
    m.args.beta = b
    m(x[:100000], return_device=True)
torch.cuda.synchronize()
t_each = (time.perf_counter() - t0) * (B / 100000)
from oracle import range_oracle as O     # checker only
obank = O.prep_bank(locs, vals, keys)
idx = np.sort(np.random.default_rng(3).choice(B, 32, replace=False))
sel = torch.from_numpy(idx).to("cuda:0")
err = 0.0
for j, b in enumerate(betas):
    got = sw[j][sel].cpu().numpy()
    err = max(err, float(np.abs(got[:, :1024] - O.retrieve64(got[:, 1024:], q[idx], obank, "RANGE+", b)).max()))
print(json.dumps({"queries": B, "betas": list(betas), "sweep_s": t_sweep,
                  "embeddings_per_s": B * len(betas) / t_sweep,
                  "one_forward_per_beta_s_extrapolated": t_each,
                  "max_abs_vs_f64_oracle": err}))
