#!/usr/bin/env python3
"""What the prepared bank file (``.rbank``) is FOR, measured: wall time and peak
memory of loading ``range_db_large`` from the reference's own schema - ``np.savez(locs,
image_embeddings, satclip_embeddings)`` with the embeddings stored as float64
(range/generate_db.py:197-214: 1.03 GB for N = 100 000) - against the prepared file (float32, unit
keys, page-aligned sections: 513 MB), for the whole bank on one GPU and for one rank of an 8-way
row-sharded job (which maps only its slice of every section).

    python tools/load_time.py [--rows 100000] [--world 8] [--dir /tmp/range_load_time] [--json out.json]

Every figure comes from a FRESH process (``--probe``), so that peak RSS is that load's own:
  host_s      ``bankfile.load_any(path)`` [+ ``.rows(r0, r1)``] - what the reference does at
              range.py:78-95 (np.load of float64 arrays, casts, float32 normalisation, trigonometry), or the mmap
  upload_s    the bank into the engine (``HipEngine.set_bank``: hipMemcpy from those arrays + the device-side
              copies); without a GPU: a host copy of the same arrays (touches the same pages)
  total_s     ``load_model(...)`` end to end (GPU only; includes the checkpoint and the encoder tables)
  peak_rss_mb ``VmHWM`` of the process; rss_file_mb = pages of the mapped FILE the process touched
              (``RssFile``): for a rank of W it must stay near 1/W of the file - tests/test_host_cpu.py pins it
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def _status_mb(key: str) -> float:
    for ln in open("/proc/self/status"):
        if ln.startswith(key + ":"):
            return int(ln.split()[1]) / 1024.0
    return float("nan")


def write_reference_npz(path: str, n_rows: int, seed: int = 2024) -> str:
    """The bank as generate_db.py writes it: embeddings cast .double() before np.savez (:197-198, :212-214)."""
    import numpy as np
    from tools import synth
    locs, vals, keys = synth.make_bank(n_rows, seed)
    np.savez(path, locs=np.asarray(locs, dtype=np.float64), image_embeddings=np.asarray(vals, dtype=np.float64),
             satclip_embeddings=np.asarray(keys, dtype=np.float64))
    return path


def probe(path: str, rank: int, world: int, ckpt: str | None) -> dict:
    import numpy as np
    import torch
    from range_amd.bankfile import load_any
    from range_amd.dist import shard_rows
    base_file = _status_mb("RssFile")
    gpu = torch.cuda.is_available()
    if gpu:
        torch.zeros(1, device="cuda:0")          # (context creation is not the bank's cost)
        torch.cuda.synchronize()
    rss0 = _status_mb("VmRSS")
    base_file = _status_mb("RssFile")
    t0 = time.perf_counter()
    bank = load_any(path)
    n = bank.n_rows
    if world > 1:
        r0, r1 = shard_rows(n, world, rank)
        bank = bank.rows(r0, r1)
    t1 = time.perf_counter()
    if gpu:
        from range_amd import _native
        eng = _native.HipEngine("cuda:0")
        eng.set_bank(bank.keys, bank.values, bank.xyz, 0)
        torch.cuda.synchronize()
    else:
        held = [np.array(a, dtype=np.float32, copy=True) for a in (bank.keys, bank.values, bank.xyz)]   # noqa: F841
    t2 = time.perf_counter()
    res = {"path": os.path.basename(path), "file_mb": os.path.getsize(path) / 2 ** 20, "bank_rows": n, "rank": rank, "world": world,
           "rows_loaded": bank.n_rows, "host_s": t1 - t0, "upload_s": t2 - t1, "gpu": gpu,
           # (VmHWM: this process image's own high-water mark; ru_maxrss would carry over the launcher's)
           "peak_rss_mb": _status_mb("VmHWM"), "rss_before_mb": rss0,
           "rss_file_mb": _status_mb("RssFile") - base_file}
    if gpu and ckpt and world == 1:
        del eng
        from range_amd import load_model
        t0 = time.perf_counter()
        m = load_model("RANGE+", pretrained_path=ckpt, device="cuda:0", db_path=path, beta=0.5)
        torch.cuda.synchronize()
        res["total_s"] = time.perf_counter() - t0
        del m
    return res


def run_probe(path, rank, world, ckpt, drop_cache=False):
    cmd = [sys.executable, os.path.abspath(__file__), "--probe", path, str(rank), str(world)]
    if ckpt:
        cmd += ["--ckpt", ckpt]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if p.returncode:
        raise SystemExit(p.stdout + p.stderr)
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--dir", default="/tmp/range_load_time")
    ap.add_argument("--json", default=None)
    ap.add_argument("--probe", nargs=3, default=None, metavar=("PATH", "RANK", "WORLD"))
    ap.add_argument("--ckpt", default=None)
    a = ap.parse_args()
    if a.probe:
        print(json.dumps(probe(a.probe[0], int(a.probe[1]), int(a.probe[2]), a.ckpt)))
        return
    from tools import synth
    from range_amd.bankfile import convert_npz
    os.makedirs(a.dir, exist_ok=True)
    npz = os.path.join(a.dir, f"range_db_{a.rows}.npz")
    rbank = os.path.join(a.dir, f"range_db_{a.rows}.rbank")
    t0 = time.perf_counter()
    if not os.path.exists(npz):
        write_reference_npz(npz, a.rows)
    t1 = time.perf_counter()
    convert_npz(npz, rbank)
    t_conv = time.perf_counter() - t1
    ck = synth.write_checkpoint(os.path.join(a.dir, "enc.ckpt"), L=40, hidden=512, seed=1234)
    out = {"bank_rows": a.rows, "npz_mb": os.path.getsize(npz) / 2 ** 20, "rbank_mb": os.path.getsize(rbank) / 2 ** 20,
           "convert_once_s": t_conv, "runs": []}
    for path in (npz, rbank):
        for rank, world in ((0, 1), (a.world // 2, a.world)):
            run_probe(path, rank, world, ck)                   # (first run: the file enters the page cache)
            r = run_probe(path, rank, world, ck)
            out["runs"].append(r)
            print(f"{r['path']:>26} rank {rank} of {world}: host {r['host_s'] * 1e3:8.1f} ms  upload {r['upload_s'] * 1e3:8.1f} ms"
                  + (f"  load_model {r['total_s'] * 1e3:8.1f} ms" if "total_s" in r else "")
                  + f"  peak RSS {r['peak_rss_mb']:7.0f} MB (before the load {r['rss_before_mb']:.0f})  file pages touched {r['rss_file_mb']:6.1f} MB"
                  f"  of {r['file_mb']:.0f} MiB", flush=True)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
