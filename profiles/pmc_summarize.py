#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per range_hip kernel (averages per
dispatch).  Usage: pmc_summarize.py <dir-with-*_counter_collection.csv> [...]"""
import collections, csv, glob, json, sys

def summarize(dirs):
    out = {}
    for d in dirs:
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(dict))
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "range_hip::" not in k:
                    continue
                k = k.split("range_hip::")[1].split("(")[0]
                agg[k][r["Counter_Name"]][r["Dispatch_Id"]] = float(r["Counter_Value"])
                agg[k]["duration_ns"][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            for k, v in agg.items():
                o = out.setdefault(k, {})
                for c, per in v.items():
                    o[c] = sum(per.values()) / len(per)
    return out

if __name__ == "__main__":
    print(json.dumps(summarize(sys.argv[1:]), indent=1))
