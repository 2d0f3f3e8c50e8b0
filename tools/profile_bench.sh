#!/bin/bash
# The rocprofv3 passes behind profiles/rNN (run on the GPU box from the repo root):
#   tools/profile_bench.sh r03
# bench line; kernel-trace stats of the same workload; three separate --pmc passes (SQ/GRBM,
# FETCH_SIZE, WRITE_SIZE+TCC) as MI355X_MICROARCH.md prescribes (never combined with other trace
# domains); the top-k scan per configuration (N = 100 000 and the DRAM-resident N = 1 000 000; 16 / 64
# queries; bf16-key product path and float32 keys): kernel trace + a separate FETCH_SIZE pass; the
# small-batch forward (tools/latency.py) kernel trace.
# Everything lands under gpurun_out/prof_<tag>/; copy the summaries into profiles/<tag>/ afterwards.
# A second argument selects a part (a box call is limited to 20 minutes): a = kernel trace, counters, scan,
# bench line; b = latency, batch top-k, rank emulations, clock ramp, encoder, load times.  Default: both.
set -e
tag=${1:-r03}
part=${2:-ab}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [[ $part == *a* ]]; then
# (--no-extras: only the timed workload, so that per-kernel averages are those of the bench geometry)
B="python3 $R/bench.py --cpu-sample 0 --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- $B --steps 10 --warmup 3 > $O/ks.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc1 -o p -- $B --steps 3 --warmup 1 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc2 -o p -- $B --steps 3 --warmup 1 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc3 -o p -- $B --steps 3 --warmup 1 > $O/p3.log 2>&1
python3 $R/profiles/pmc_summarize.py $O/pmc1 $O/pmc2 $O/pmc3 > $O/pmc_summary.json
# the traffic figures bench.py reports, stamped with the hash of the kernel sources measured (copy to profiles/attend_pmc.json)
python3 $R/bench.py --no-extras --cpu-sample 0 2> /dev/null | tail -1 > $O/bench_line_noextras.json
python3 $R/profiles/make_attend_pmc.py $O/pmc_summary.json $O/bench_line_noextras.json $tag > $O/attend_pmc.json
mkdir -p $O/scan
for n in 100000 1000000; do for q in 16 32 64; do for k in bf16 f32; do
  rocprofv3 --kernel-trace --output-format csv -d $O/scan/ks_${n}_${q}_${k} -o t -- python3 $R/tools/scan_bench.py --n $n --q $q --keys $k > $O/scan/ks_${n}_${q}_${k}.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/scan/pmc_${n}_${q}_${k} -o p -- python3 $R/tools/scan_bench.py --n $n --q $q --keys $k > $O/scan/pmc_${n}_${q}_${k}.log 2>&1
  echo "scan $n $q $k done"
done; done; done
# (stamped with the hash of the kernel sources measured; bench.py quotes the medians while the stamp is this checkout's)
python3 $R/profiles/scan_summarize.py $O/scan $(python3 $R/range_amd/_srchash.py) > $O/scan_summary.json
cp $O/scan_summary.json $R/profiles/scan_summary.json
# the bench line proper, with the traffic of the passes above (stamped with this checkout's source hash)
cp $O/attend_pmc.json $R/profiles/attend_pmc.json
(cd $R && python3 bench.py > $O/bench.log 2> $O/bench.err)
tail -1 $O/bench.log > $O/bench_line.json
find $O/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
echo "bench passes done"
fi
if [[ $part == *b* ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat -o lat -- python3 $R/tools/latency.py > $O/latency.log 2>&1
find $O/lat -name "*kernel_stats.csv" -exec cp {} $O/latency_kernel_stats.csv \;
python3 $R/tools/latency.py > $O/latency_plain.log 2>&1
# round 5: the batch-scale top-k (kernel trace), what a rank of 8 computes (steady state / the old cold
# protocol / layouts), the clock ramp behind the difference, the mid-size encoder A/B
rocprofv3 --kernel-trace --stats --output-format csv -d $O/topk -o tk -- python3 $R/tools/topk_batch.py 10000 100000 > $O/topk_batch.log 2>&1
find $O/topk -name "*kernel_stats.csv" -exec cp {} $O/topk_batch_kernel_stats.csv \;
python3 $R/tools/shard_emulate.py 1 2 4 8 > $O/shard_emulate_steady.log 2>&1
python3 $R/tools/shard_emulate.py --chunks 1 8 > $O/shard_emulate_one_chunk.log 2>&1
python3 $R/tools/shard_emulate.py --cold 8 > $O/shard_emulate_cold_protocol.log 2>&1
python3 $R/tools/shard_emulate.py --layouts 8 > $O/shard_emulate_layouts.log 2>&1
python3 $R/tools/clock_ramp.py 10000 12500 300 > $O/clock_ramp.log 2>&1
python3 $R/tools/clock_ramp.py 10000 12500 100 3 >> $O/clock_ramp.log 2>&1
python3 $R/tools/encoder_mid.py > $O/encoder_mid.log 2>&1
# round 6: what the prepared bank file is for (reference-schema float64 npz against .rbank, whole bank / a rank of 8)
python3 $R/tools/load_time.py --dir /tmp/range_load_time --json $O/load_time.json > $O/load_time.log 2>&1
rm -rf /tmp/range_load_time
fi
[ -f $O/bench_line.json ] && cut -c1-300 $O/bench_line.json
ls $O
