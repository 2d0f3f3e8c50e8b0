#!/usr/bin/env python3
"""The single-GPU configurations of BASELINE.json at full size (synthetic banks / weights, see
SURVEY.md 8(d)), one JSON line each:
  C2  RANGE+ beta=0.5, range_db_med (N=50 000), 10 000 queries: parity of a query sample against the
      float64 oracle and the reference's float32 op order, top-16 indices against the oracle
  C3  RANGE+ beta=0.5, range_db_large (N=100 000), 100 000 queries through the Python API (chunks of
      16 384): throughput, device-resident output, and parity of a sample
  C5' beta sweep {0, .25, .5, .75, 1} on range_db_large, 100 000 queries on one GPU
The oracle is used as the checker only."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import range_oracle as O
from range_amd import load_model, synth

tmp = tempfile.mkdtemp()
L, H = 40, 512
ck = synth.write_checkpoint(os.path.join(tmp, "e.ckpt"), L=L, hidden=H, seed=1234)
w = synth.make_encoder_weights(L, H, 256, 2, 1234)


def check(model, q, bank, beta, n=192):
    idx = np.random.default_rng(1).choice(q.shape[0], n, replace=False)
    out = model(torch.from_numpy(q[idx]).cuda())
    e = O.encode(q[idx], w, L)
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    ref64 = O.retrieve64(e, q[idx], bank, "RANGE+", beta)
    ref32 = O.retrieve(e, q[idx], bank, "RANGE+", beta)
    return {"max_abs_vs_f64_oracle": float(np.abs(out[:, :1024] - ref64).max()),
            "max_abs_vs_reference_f32_order": float(np.abs(out - ref32).max()),
            "ehat_max_abs": float(np.abs(out[:, 1024:] - e).max())}, idx, e


for name, N, B in (("C2 range_db_med 10k queries", synth.BANK_ROWS["range_db_med"], 10_000),
                   ("C3 range_db_large 100k queries", synth.BANK_ROWS["range_db_large"], 100_000)):
    db = synth.write_bank(os.path.join(tmp, f"db{N}.npz"), N, 2024)
    bank = O.load_bank(db)
    m = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db, beta=0.5)
    q = synth.make_queries(B, seed=7)
    x = torch.from_numpy(q).cuda()
    m(x[:1000], return_device=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m(x, return_device=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    par, idx, e = check(m, q, bank, 0.5)
    tv, ti = m.topk(x[torch.from_numpy(idx).cuda()], 16)
    s64 = e @ bank.keys.astype(np.float64).T
    rv, ri = O.topk64(s64, 16)
    res = {"config": name, "bank_rows": N, "queries": B, "seconds": dt, "geo_embeddings_per_s": B / dt,
           **par, "topk16_index_mismatches": int((ti.cpu().numpy() != ri).sum()),
           "topk16_value_max_abs": float(np.abs(tv.cpu().numpy() - rv).max())}
    print(json.dumps(res), flush=True)
    if N == synth.BANK_ROWS["range_db_large"]:
        betas = (0.0, 0.25, 0.5, 0.75, 1.0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sw = m.sweep(x, betas, return_device=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        j = 3
        mb = load_model("RANGE+", pretrained_path=ck, device="cuda:0", db_path=db, beta=betas[j])
        d = float((sw[j, :2000] - mb(x[:2000], return_device=True)).abs().max())
        print(json.dumps({"config": "C5' beta sweep on one GPU", "bank_rows": N, "queries": B,
                          "betas": betas, "seconds": dt,
                          "geo_embeddings_per_s_all_betas": len(betas) * B / dt,
                          "max_abs_vs_forward_beta0.75": d}), flush=True)
    del m
