#!/bin/bash
# The N > 1 code path of bench.py on the one-GPU box: rendezvous, RCCL communicator (world of one rank), reductions, gather of
# detections -- all through the library (aev_comm_*), no PyTorch in the process.  -> profiles/r04_bench_force_dist_one_rank.json
set -e
mkdir -p gpurun_out/r4dist
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
  python bench.py --force-dist --steps 20 --warmup 5 --no-single --no-host-feed --no-dense --no-legacy --no-cpu-baseline \
  > gpurun_out/r4dist/bench_dist1.json 2> gpurun_out/r4dist/bench_dist1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4dist/bench_dist1.json"))
print("dist backend", d["config"]["dist_backend"], "| value %.2f M events/s" % (d["value"] / 1e6), "| runtime", d["runtime"])
assert d["config"]["dist_backend"] == "rccl" and d["runtime"]["matched"]
PY
