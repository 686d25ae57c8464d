R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s1t; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --streams 1 --steps 24 --warmup 8 --profile-steps 0 --sustained-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy > $O/log.txt 2>&1
cd $R; python tools/timeline.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 7 2 > $O/timeline_S1.txt; rm -rf $O/tr
