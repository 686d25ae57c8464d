#!/bin/bash
# The artefacts under profiles/ (one box, no torch in the process): bench line (cold start / sustained / headline), BASELINE configs 3 and
# 5, batch sweep, one stream, rocprofv3 stats (graph replays + the eager per-kernel pass), PMC traffic / L2 passes, timelines.
#   usage: tools/runs/r5_profiles.sh <tag>     -> gpurun_out/<tag>/ ; copy what is to be judged into profiles/<tag>_*
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=${1:-r05}; O=gpurun_out/$T; mkdir -p $O
Q="--no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6"
python bench.py > $O/bench_S256.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --kind dense1M --batch 200 $Q --sustained-steps 0 > $O/config3_dense1M_B200.json 2>/dev/null
python bench.py --kind dense1M --batch 1000 $Q --sustained-steps 0 > $O/config3_dense1M_B1000.json 2>/dev/null
python bench.py --dense $Q --sustained-steps 0 > $O/config5_dense_frame_mode.json 2>/dev/null
python bench.py --sweep-batch > $O/batch_sweep.json 2>/dev/null; echo "sweep rc=$?"
python bench.py --streams 1 --steps 400 --warmup 20 --profile-steps 0 --sustained-steps 0 $Q > $O/bench_S1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
P="--sustained-steps 0 $Q"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --steps 24 --warmup 8 --profile-steps 0 $P > $R/$O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_eager -- python3 $R/bench.py --steps 1 --warmup 1 --profile-steps 6 $P > $R/$O/bench_eager_pass_S256.json 2> $R/$O/stats_eager.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats1 -- python3 $R/bench.py --streams 1 --steps 24 --warmup 8 --profile-steps 0 $P > $R/$O/stats1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --profile-steps 0 $P --no-graph > $R/$O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --profile-steps 0 $P --no-graph > $R/$O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/$O/pmc_l2 -- python3 $R/bench.py --steps 6 --warmup 2 --profile-steps 0 $P --no-graph > $R/$O/pmc_l2.log 2>&1
cd $R
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_S256.csv
cp $(find $O/stats_eager -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_eager_pass_S256.csv
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_S1.csv
python tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 7 4 > $O/timeline_S256.txt
python tools/timeline.py $(find $O/stats1 -name "*kernel_trace.csv" | head -1) 7 2 > $O/timeline_S1.txt
for f in $O/timeline_S256.txt $O/timeline_S1.txt; do [ -s $f ] || { echo "EMPTY ARTEFACT: $f"; exit 1; }; done
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic_S256.json 256 ncaltech $(find $O/pmc_l2 -name "*counter_collection.csv" | head -1) | tee $O/pmc_summary.txt | head -20
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_l2 $O/stats $O/stats_eager $O/stats1
python tools/show_bench.py $O/bench_S256.json > $O/bench_summary.txt; head -3 $O/bench_summary.txt
ls -la $O
