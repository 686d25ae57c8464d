#!/bin/bash
# round 2: the artefacts under profiles/ (bench line, rocprofv3 stats, PMC passes, timeline, BASELINE configs 3 and 5)
T=${1:-r02}
O=gpurun_out/$T; mkdir -p $O
python bench.py > $O/bench_S256.json 2> $O/bench.err
python bench.py --kind dense1M --batch 200 --no-cpu-baseline --no-single --no-host-feed > $O/config3_dense1M_B200.json 2>/dev/null
python bench.py --kind dense1M --batch 1000 --no-cpu-baseline --no-single --no-host-feed > $O/config3_dense1M_B1000.json 2>/dev/null
python bench.py --dense --no-cpu-baseline --no-single --no-host-feed > $O/config5_dense_frame_mode.json 2>/dev/null
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $R/$O/stats.log 2>&1
# the per-kernel pass of bench.py alone (6 eager steps on one stream, which is what bench.py's roofline block times) + 2 graph steps
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_eager -- python3 $R/bench.py --steps 1 --warmup 1 --profile-steps 6 --no-cpu-baseline --no-single --no-host-feed > $R/$O/stats_eager.json 2> $R/$O/stats_eager.log
cp $(find $R/$O/stats_eager -name "*kernel_stats.csv" | head -1) $R/$O/rocprofv3_kernel_stats_eager_pass_S256.csv; rm -rf $R/$O/stats_eager
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-graph > $R/$O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-graph > $R/$O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/$O/pmc_l2 -- python3 $R/bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-graph > $R/$O/pmc_l2.log 2>&1
cd $R
python tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 3 > $O/timeline_S256.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_S256.csv
python tools/pmc_summary.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic_S256.json 256 ncaltech $(find $O/pmc_l2 -name "*counter_collection.csv" | head -1) | tee $O/pmc_summary.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_l2 $O/stats
python tools/show_bench.py $O/bench_S256.json > $O/bench_summary.txt; head -3 $O/bench_summary.txt
