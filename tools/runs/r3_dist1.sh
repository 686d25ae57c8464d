#!/bin/bash
# rehearsal of the N > 1 code path on one GPU: torch first (its bundled HIP runtime), RCCL process group of one rank, reductions, all-gather
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/r3dist; mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?"; tail -5 $O/bench_dist1.err
python -c "
import json; d=json.load(open('$O/bench_dist1.json')); print(d['value'], d['ms_per_step'], d['runtime'], d['config']['detect_and_gather_ms'])"
