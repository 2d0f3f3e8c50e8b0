#!/bin/bash
# The ceilings the roofline fractions are argued against (round-5 verdict, item 3), measured and kept:
#   tools/ceilings.sh r06      (GPU box, from the repo root; results under gpurun_out/prof_<tag>/ceilings/)
# tools/micro/mfma_f64_peak.hip - the v_mfma_f64_16x16x4_f64 rate an MFMA-only loop sustains (DESIGN 3.1:
# the encoder's practical ceiling) - and tools/micro/mfma_f16_peak.hip - the 16-bit MFMA rate on constant
# against random operands (the batch top-k's) - each run plain (its own HIP-event lines) and
# under rocprofv3 --kernel-trace --stats (the program directly behind --), whose per-kernel durations must
# tell the same story.
set -e
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag/ceilings
mkdir -p $O
for m in mfma_f64_peak mfma_f16_peak; do
  hipcc -w --offload-arch=gfx950 -O3 -o $O/$m $R/tools/micro/$m.hip
done
cd /tmp && export TMPDIR=/tmp
for m in mfma_f64_peak mfma_f16_peak; do
  $O/$m > $O/$m.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$m -o ks -- $O/$m > $O/${m}_rocprof.log 2>&1
  find $O/ks_$m -name "*kernel_stats.csv" -exec cp {} $O/${m}_kernel_stats.csv \;
  find $O/ks_$m -name "*kernel_trace.csv" -exec cp {} $O/${m}_kernel_trace.csv \;
  rm -rf $O/ks_$m $O/$m
done
cat $O/mfma_f64_peak.log $O/mfma_f16_peak.log
