#!/bin/bash
# The rocprofv3 passes behind profiles/rNN (run on the GPU box from the repo root):
#   tools/profile_bench.sh r02
# bench line, kernel-trace stats of the same command, three separate --pmc passes (SQ/GRBM,
# FETCH_SIZE, WRITE_SIZE+TCC) as MI355X_MICROARCH.md prescribes, the small-batch top-k scan, and
# the two arithmetic modes of pass 2 side by side (tools/pv_modes.py).
# Everything lands under gpurun_out/prof_<tag>/; copy the summaries into profiles/<tag>/ afterwards
# (profiles/pmc_summarize.py condenses the counter files).
set -e
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
python $R/bench.py > $O/bench.log 2> $O/bench.err
tail -1 $O/bench.log > $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
# (--no-extras: only the timed workload, so that per-kernel averages are those of the bench geometry)
B="python3 $R/bench.py --cpu-sample 0 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- $B --steps 10 --warmup 3 > $O/ks.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc1 -o p -- $B --steps 3 --warmup 1 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc2 -o p -- $B --steps 3 --warmup 1 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc3 -o p -- $B --steps 3 --warmup 1 > $O/p3.log 2>&1
RANGE_TOPKS_KEYS=f32 RANGE_TOPKS_GROUPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scan1 -o ts -- python3 $R/tools/small_batch_scan.py --stream-only > $O/scan1.log 2>&1
RANGE_TOPKS_KEYS=f32 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scanf -o ts -- python3 $R/tools/small_batch_scan.py --stream-only > $O/scanf.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/scan_pmc -o p -- python3 $R/tools/small_batch_scan.py --stream-only > $O/scan_pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/scan -o ts -- python3 $R/tools/small_batch_scan.py --stream-only > $O/scan.log 2>&1
RANGE_TOPKS_KEYS=f32 RANGE_TOPKS_GROUPS=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/scan1_pmc -o p -- python3 $R/tools/small_batch_scan.py --stream-only > $O/scan1_pmc.log 2>&1
# the opt-in pass 2 on bf16 planes next to the exact one: times + error vs the float64 oracle, kernel trace, fetched bytes
python3 $R/tools/pv_modes.py --json > $O/pv_modes.json 2> $O/pv_modes.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pv -o pv -- python3 $R/tools/pv_modes.py > $O/pv.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pv_pmc -o p -- python3 $R/tools/pv_modes.py > $O/pv_pmc.log 2>&1
find $O/pv -name "*kernel_stats.csv" -exec cp {} $O/pv_modes_kernel_stats.csv \;
python3 $R/profiles/pmc_summarize.py $O/pmc1 $O/pmc2 $O/pmc3 > $O/pmc_summary.json
cp $O/ks/ks_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || find $O/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
cut -c1-300 $O/bench_line.json
ls $O
