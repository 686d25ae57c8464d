"""CPU, world_size 2 over gloo: the only multi-rank step of the path -- sharding independent streams over ranks and
all-gathering the per-stream predictions -- exercised exactly as bench.py does it on N GPUs (there over RCCL)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from async_ev_cnn_amd.streams import shard_streams, gather_predictions
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# 6 streams shard evenly over 2 ranks; 7 do not (rank 0 holds 4, rank 1 holds 3): the gather pads and returns the counts
for total in (6, 7):
    mine = shard_streams(total, rank, world)
    assert mine == list(range(rank, total, world))
    # stand-in predictions: stream i -> tensor filled with i + 1 (shape of the e-fcn output [5,7,110])
    local = torch.stack([torch.full((5, 7, 110), float(i + 1)) for i in mine])
    allp, counts = gather_predictions(local)
    shards = [shard_streams(total, r, world) for r in range(world)]
    assert counts == [len(sh) for sh in shards]
    assert allp.shape == (world, max(counts), 5, 7, 110)
    for r in range(world):
        for j, i in enumerate(shards[r]):
            assert float(allp[r, j, 0, 0, 0]) == float(i + 1)
        assert float(allp[r, counts[r]:].abs().sum()) == 0.0      # padding rows
# the bench's max-over-ranks timing reduction
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == float(world)
ev = torch.tensor([100.0 * (rank + 1)], dtype=torch.float64)
dist.all_reduce(ev, op=dist.ReduceOp.SUM)
assert float(ev) == 100.0 * world * (world + 1) / 2
dist.barrier()
if rank == 0:
    print("DIST_OK")
dist.destroy_process_group()
'''


def test_shard_and_gather_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "DIST_OK" in out.stdout
