#!/bin/bash
# Round 3: the one-stream artefacts under profiles/ again (after a change of the few-streams plan) + the streams-per-GPU sweep
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=${1:-r03f}; O=gpurun_out/${T}_s1; mkdir -p $O
Q="--no-cpu-baseline --no-single --no-host-feed --no-dense"
python bench.py --streams 1 --steps 400 --warmup 20 --profile-steps 0 $Q > $O/bench_S1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats1 -- python3 $R/bench.py --streams 1 --steps 24 --warmup 8 --profile-steps 0 $Q > $R/$O/stats1.log 2>&1
cd $R
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_S1.csv
python tools/timeline.py $(find $O/stats1 -name "*kernel_trace.csv" | head -1) 7 2 > $O/timeline_S1.txt
rm -rf $O/stats1
tools/scale_sweep.sh > $O/streams_per_gpu_sweep.txt 2>&1
cat $O/bench_S1.json | head -c 400; echo; cat $O/streams_per_gpu_sweep.txt
