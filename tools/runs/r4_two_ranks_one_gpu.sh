#!/bin/bash
# Two ranks on the ONE GPU of the box, launched as the driver launches N ranks: RCCL refuses two ranks on one device, so every rank's
# communicator fails, the ranks agree on that over the TCP star and keep it (comm.open_comm) -- the N > 1 control flow and the fallback
# transport end to end on hardware.  -> profiles/r04_two_ranks_one_gpu_tcp_fallback.json
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 \
  --steps 10 --warmup 3 --sustained-steps 20 --streams 64 --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy \
  > gpurun_out/two_ranks_one_gpu.json 2> gpurun_out/two_ranks_one_gpu.err
python - <<'PY'
import json
lines = [x for x in open("gpurun_out/two_ranks_one_gpu.json") if x.startswith("{")]
d = json.loads(lines[-1])
assert len(lines) == 1 and d["n_gpus"] == 2 and d["config"]["dist_backend"].startswith("tcp: ") and d["runtime"]["matched"]
print("one JSON line, n_gpus 2, %.1f M events/s, dist_backend: %s" % (d["value"] / 1e6, d["config"]["dist_backend"][:100]))
PY
