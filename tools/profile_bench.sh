#!/bin/bash
# The rocprofv3 passes behind profiles/rNN (run on the GPU box from the repo root):
# bench line, kernel-trace stats, three separate --pmc passes (SQ/GRBM, FETCH_SIZE, WRITE_SIZE+TCC).
set -e
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01c -o ks -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-sample 0 > $R/gpurun_out/prof_r01c_ks.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/prof_r01c/pmc1 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof_r01c_p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01c/pmc2 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof_r01c_p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof_r01c/pmc3 -o p -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof_r01c_p3.log 2>&1
tail -1 $R/gpurun_out/bench.log | cut -c1-200
ls $R/gpurun_out/prof_r01c
