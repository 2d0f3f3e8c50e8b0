#!/usr/bin/env python3
"""Per-configuration summary of the top-k scan's rocprofv3 passes (tools/profile_bench.sh):
  scan_summarize.py <dir>   with sub-directories  ks_<N>_<q>_<keys>/ (--kernel-trace) and
                            pmc_<N>_<q>_<keys>/ (--pmc FETCH_SIZE, a separate pass)
For every configuration: median / mean duration of the stream kernel (which carries the merge as
its tail: one launch per call up to 256 queries) over all its dispatches in the trace, the bytes
the call streams, the fraction of 8 TB/s, and FETCH_SIZE x 2 (the gfx950 correction of
MI355X_MICROARCH.md: a wide coalesced stream is tallied at half its bytes) per dispatch."""
import csv, glob, json, os, re, statistics, sys


def durations(d, pat):
    out = []
    for f in glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if re.search(pat, r["Kernel_Name"]):
                out.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out


def fetch(d, pat):
    per = {}
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if re.search(pat, r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE":
                per[r["Dispatch_Id"]] = float(r["Counter_Value"])
    return list(per.values())


def main(root):
    rows = []
    for ks in sorted(glob.glob(f"{root}/ks_*")):
        m = re.match(r"ks_(\d+)_(\d+)_(\w+)", os.path.basename(ks))
        n, q, keys = int(m.group(1)), int(m.group(2)), m.group(3)
        pat = "topk_stream_bf16_kernel" if keys == "bf16" else r"topk_stream_kernel"
        du = durations(ks, pat)
        if not du:
            continue
        groups = (q + 15) // 16
        per_pass = 1 if groups <= 1 else 2
        passes = (groups + per_pass - 1) // per_pass
        streamed = passes * n * (512 if keys == "bf16" else 1024)
        med = statistics.median(du)
        fs = fetch(f"{root}/pmc_{n}_{q}_{keys}", pat)
        rows.append({"bank_rows": n, "queries": q, "keys": keys, "passes": passes, "dispatches": len(du),
                     "kernel_median_ns": med, "kernel_mean_ns": round(statistics.mean(du)),
                     "streamed_bytes": streamed, "streamed_TBps_median": round(streamed / med / 1e3, 3),
                     "frac_of_8TBps_median": round(streamed / med / 1e3 / 8.0, 4),
                     "frac_of_8TBps_mean": round(streamed / statistics.mean(du) / 1e3 / 8.0, 4),
                     "fetch_size_x2_bytes_per_dispatch": round(2 * 1024 * statistics.mean(fs)) if fs else None,
                     "resident": "dram" if n * (512 if keys == "bf16" else 1024) > 256e6 else "infinity_cache"})
    print(json.dumps({"csrc_sha256": sys.argv[2] if len(sys.argv) > 2 else None,
                      "what": "rocprofv3 --kernel-trace of tools/scan_bench.py per configuration (3 warm-up + 60 timed calls; "
                              "each call = ONE launch: the stream kernel with the merge as its tail); FETCH_SIZE from a "
                              "separate --pmc pass of the same command, x2 (gfx950) and in bytes (the counter is in KB)",
                      "runs": rows}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
