"""CPU, world_size 2 over gloo: the only multi-rank step of the path -- sharding independent streams over ranks and
all-gathering the per-stream predictions -- exercised exactly as bench.py does it on N GPUs (there over RCCL)."""
import json
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


BENCH_WORKER = r'''
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
import bench


class StubNet:
    """Stands in for EventNet on a CPU rank: counts what it is asked to do, produces recognisable detections."""
    shapes = [[1, 160, 224], [110, 5, 7]]
    names, kinds, conv_specs = ["intgr", "conv7"], ["intgr", "conv"], [None, (1, 1, 512, 110)]

    def __init__(self, n_streams, dev):
        self.n_streams, self.dev, self.ran, self.events = n_streams, dev, 0, 0

    def set_conv_impl(self, impl): pass
    def set_dense(self, d): pass
    def queue_upload(self, steps):
        assert all(len(st) == self.n_streams for st in steps)
        self.steps = steps
    def queue_run(self, t0, T, graph):
        self.ran += T
        self.events += sum(len(b) for t in range(t0, t0 + T) for b in self.steps[t])
    def sync(self): pass
    def stats(self, reset=False):
        r = (np.array([self.events, self.events]), self.events, self.ran)
        if reset:
            self.events = 0
        return r


class StubEngine:
    def device_count(self): return 8
    def device_sync(self, dev): pass
    def runtime_info(self): return {"stub": True}
    def select_device(self, torch, dev): pass
    def make_net(self, n_streams, dev, tuning="auto"):
        assert dev == int(os.environ["LOCAL_RANK"])
        return StubNet(n_streams, dev)
    def detections(self, net, torch, on_device):
        assert not on_device
        rank = int(os.environ["RANK"])
        return torch.full((net.n_streams, 70, 7), float(rank + 1))


line = bench.main(["--gpus", "8", "--steps", "4", "--warmup", "1", "--streams", "3", "--profile-steps", "0", "--dist-backend", "gloo"],
                  engine=StubEngine())
rank = int(os.environ["RANK"])
assert (line is not None) == (rank == 0)            # only rank 0 prints / returns the line
if rank == 0:
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["streams_per_gpu"] == 3
    # value = events of ALL ranks' streams over the MAX time: 8 ranks x 3 streams x 4 timed steps x ~200 events
    ev = line["value"] * line["ms_per_step"] * 1e-3 * 4
    assert 8 * 3 * 4 * 150 < ev < 8 * 3 * 4 * 260, ev
    print("BENCH8_OK")
'''


def test_bench_control_flow_world8_gloo_with_stub_engine(tmp_path):
    """bench.main's multi-rank control flow at the driver's largest size (8 ranks), on CPU ranks over gloo with a stub in place of
    the GPU engine: stream sharding (24 global streams, stream i -> rank i mod 8), barriers, the MAX / SUM reductions, the gather
    of detections, rank-0-only printing.  What stays unproven without an 8-GPU node is RCCL itself."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", "29619", str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "BENCH8_OK" in out.stdout
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 8      # ONE JSON line, from rank 0
