#!/usr/bin/env python3
"""profiles/attend_pmc.json from a round's PMC passes (tools/profile_bench.sh): the HBM bytes per launch of
the dominant kernel and of every kernel of the step, corrected as MI355X_MICROARCH.md prescribes for
gfx950 (FETCH_SIZE counts 128-byte requests as 64: x2; WRITE_SIZE as read; both in KB), stamped with the
SHA-256 of the kernel sources they were measured on.  bench.py reports `roofline.traffic` only while the
sources still hash to that stamp (a figure measured on other kernels is not this run's traffic).

Usage: make_attend_pmc.py <pmc_summary.json> <bench_line.json> <round tag> > profiles/attend_pmc.json"""
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SURVEY_HBM_MINIMAL_BYTES = 100_000 * 5132 + 10_000 * (16 + 10_240)     # one bank pass + per-query I/O


def csrc_sha256():
    """the hash build.sh embeds in the library and bench.py compares against (range_amd/_srchash.py)"""
    sys.path.insert(0, REPO)
    from range_amd._srchash import source_sha256
    return source_sha256()


def hbm_bytes(k):
    return k.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0, k.get("WRITE_SIZE", 0.0) * 1024.0


def main():
    pmc = json.load(open(sys.argv[1]))
    line = json.load(open(sys.argv[2]))
    tag = sys.argv[3] if len(sys.argv) > 3 else "?"
    cfg, roof = line["config"], line["roofline"]
    stored = "attend_stored_kernel" in roof["kernel"]
    key = "attend_stored_kernel<true>" if stored else "attend_kernel<true>"
    k = pmc[key]
    rd, wr = hbm_bytes(k)
    # every kernel of one step of the bench workload: launches per step x bytes per launch
    step = {}
    for name, per_step in (("encoder_kernel<8, 16>", 1), ("encoder_l1_part_kernel<8, 16>", 1), ("encoder_l2_part_kernel<4>", 1),
                           ("encoder_rest_kernel<8, 16>", 1), ("scan_stats_kernel<true, false>", 1), ("merge_stats_kernel", 1),
                           (key, 1), ("finalize_kernel", 1)):
        if name in pmc:
            r, w = hbm_bytes(pmc[name])
            step[name] = {"hbm_read_bytes": r * per_step, "hbm_write_bytes": w * per_step}
    total = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in step.values())
    entry = {
        "kernel_key": key, "queries": roof["queries_per_launch"], "bank_rows": cfg["bank_rows_per_gpu"],
        "query_tiles": cfg["query_tiles"], "bank_splits": cfg["bank_splits"],
        "workload": "bench.py default (RANGE+, B=10000, N=100000, kept logits)",
        "FETCH_SIZE_KB_raw": k.get("FETCH_SIZE"), "WRITE_SIZE_KB_raw": k.get("WRITE_SIZE"),
        "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B); WRITE_SIZE as read (MI355X_MICROARCH.md, HBM)",
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
        "l2_hit_rate": (k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])) if "TCC_HIT_sum" in k else None,
        "mfma_busy_frac": (k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_BUSY_CYCLES"] / 32.0) if "SQ_BUSY_CYCLES" in k and k["SQ_BUSY_CYCLES"] else None,
        "effective_clock_GHz": (k["GRBM_GUI_ACTIVE"] / 8.0 / k["duration_ns"]) if "GRBM_GUI_ACTIVE" in k else None,
        "step_kernels": step, "hbm_bytes_per_step": total,
        "survey_hbm_minimal_bytes_per_step": SURVEY_HBM_MINIMAL_BYTES,
        "hbm_bytes_per_step_over_survey_minimal": total / SURVEY_HBM_MINIMAL_BYTES,
        "source": f"profiles/{tag}/pmc_summary.json (tools/profile_bench.sh {tag}: three separate rocprofv3 --pmc passes of bench.py --no-extras)",
        "csrc_sha256": csrc_sha256(),
    }
    print(json.dumps([entry], indent=1))


if __name__ == "__main__":
    main()
