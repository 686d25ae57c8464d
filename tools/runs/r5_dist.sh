#!/bin/bash
# The N > 1 code path on the one-GPU box (round 5: bring-up probed in child processes, loud keys):
#   1. a world of ONE rank through bench.py --force-dist: probe child, RCCL communicator, reductions, gather -> rccl_ranks 1, dist_backend_ok true
#   2. TWO ranks on the box's one GPU, launched as the driver launches N ranks: RCCL refuses two ranks on one device -- the probe children
#      say so, no rank ever enters RCCL in its own process, the ranks keep the TCP star -> rccl_ranks 0, dist_backend_ok false
set -e
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/r5dist; mkdir -p $O
Q="--no-single --no-host-feed --no-dense --no-legacy --no-x6 --no-cpu-baseline"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 600 python bench.py --force-dist --steps 20 --warmup 5 $Q > $O/bench_dist1.json 2> $O/bench_dist1.err
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 \
  --steps 10 --warmup 3 --sustained-steps 20 --streams 64 $Q > $O/two_ranks_one_gpu.json 2> $O/two_ranks_one_gpu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5dist/bench_dist1.json"))
print("one rank : dist backend", d["config"]["dist_backend"], "| rccl_ranks", d["rccl_ranks"], "| dist_backend_ok", d["dist_backend_ok"], "| %.2f M events/s" % (d["value"] / 1e6), "| runtime matched", d["runtime"]["matched"])
assert d["config"]["dist_backend"] == "rccl" and d["rccl_ranks"] == 1 and d["dist_backend_ok"] is True and d["runtime"]["matched"]
lines = [x for x in open("gpurun_out/r5dist/two_ranks_one_gpu.json") if x.startswith("{")]
d = json.loads(lines[-1])
print("two ranks: one JSON line %s, n_gpus %d, rccl_ranks %d, dist_backend_ok %s, %.1f M events/s, dist_backend: %s" % (len(lines) == 1, d["n_gpus"], d["rccl_ranks"], d["dist_backend_ok"], d["value"] / 1e6, d["config"]["dist_backend"][:120]))
assert len(lines) == 1 and d["n_gpus"] == 2 and d["config"]["dist_backend"].startswith("tcp: ") and d["rccl_ranks"] == 0 and d["dist_backend_ok"] is False and d["runtime"]["matched"]
PY
