# Further --pmc passes (LDS conflicts, instruction mix) of the bench workload and of the two
# arithmetic modes of pass 2: tools/pmc_extra.sh, results under gpurun_out/pmc_extra/, condensed
# per kernel by the python at the end (copy its output to profiles/<round>/pmc_extra.json).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_extra
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prog in "bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras" "tools/pv_modes.py"; do
  tag=$(echo $prog | cut -d. -f1 | tr '/' '_')
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${tag}_a -o p -- python3 $R/$prog > $O/${tag}_a.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $O/${tag}_b -o p -- python3 $R/$prog > $O/${tag}_b.log 2>&1 || true
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/${tag}_c -o p -- python3 $R/$prog > $O/${tag}_c.log 2>&1 || true
done
python3 - <<'PY'
import csv, glob, json, os, collections
O = os.path.expandvars("$GRAFT_REPO_ROOT/gpurun_out/pmc_extra")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void range_hip::", "").replace("range_hip::", "")
        if not any(s in k for s in ("encoder", "scan_stats", "attend", "finalize", "merge_stats")):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["duration_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(O + "/pmc_extra.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: round(v) for c, v in d.items() if c in ("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS")})
PY
