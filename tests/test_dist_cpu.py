"""CPU, world_size 2 over gloo: the only multi-rank step of the path -- sharding independent streams over ranks and
all-gathering the per-stream predictions -- exercised exactly as bench.py does it on N GPUs (there over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

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
    def make_net(self, n_streams, dev, tuning="auto"):
        assert dev == int(os.environ["LOCAL_RANK"])
        return StubNet(n_streams, dev)
    def detections(self, net):
        return np.full((net.n_streams, 70, 7), float(int(os.environ["RANK"]) + 1), np.float32)
    def make_comm(self, dev):
        # the product's host-side star (async_ev_cnn_amd.comm.TcpComm: what open_comm falls back to when RCCL cannot be brought up),
        # here in place of the RCCL communicator -- RCCL needs GPUs; the detections are the stub's
        from async_ev_cnn_amd import comm
        class Star(comm.TcpComm):
            def gather_detections(self, net, *a, **k):
                parts = [np.frombuffer(p, np.float32).reshape(-1, 70, 7) for p in self.allgather_bytes(StubEngine().detections(net).tobytes())]
                return np.stack(parts), [len(p) for p in parts]
        c = Star.from_env()
        if mode == "tcpfallback":   # what comm.open_comm hands back when RCCL could not be brought up on every rank
            c.backend, c.rccl_ranks = "tcp: rank 3: no RCCL in this test", 0
        else:                       # ... and when it could: aev_comm_world of the communicator
            c.rccl_ranks = c.world
        return c


mode = sys.argv[1]
backend = "rccl" if mode == "tcpfallback" else mode
line = bench.main(["--gpus", "8", "--steps", "4", "--warmup", "1", "--streams", "3", "--profile-steps", "0", "--sustained-steps", "6",
                   "--dist-backend", backend], engine=StubEngine())
rank = int(os.environ["RANK"])
assert backend != "rccl" or "torch" not in sys.modules      # the product's multi-rank path never imports PyTorch
assert (line is not None) == (rank == 0)            # only rank 0 prints / returns the line
if rank == 0:
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["streams_per_gpu"] == 3
    assert line["sustained"]["steps"] == 6
    # an RCCL failure is loud at the TOP level of the line: rccl_ranks / dist_backend_ok
    if mode == "tcpfallback":
        assert line["dist_backend_ok"] is False and line["rccl_ranks"] == 0 and line["config"]["dist_backend"].startswith("tcp: rank 3")
    else:
        assert line["dist_backend_ok"] is True and line["config"]["dist_backend"] == backend
        assert line["rccl_ranks"] == (8 if mode == "rccl" else 0)
    # value = events of ALL ranks' streams over the MAX time: 8 ranks x 3 streams x 4 timed steps x ~200 events
    ev = line["value"] * line["ms_per_step"] * 1e-3 * 4
    assert 8 * 3 * 4 * 150 < ev < 8 * 3 * 4 * 260, ev
    print("BENCH8_OK")
'''


@pytest.mark.parametrize("backend,port", [("rccl", "29631"), ("gloo", "29619"), ("tcpfallback", "29667")])
def test_bench_control_flow_world8_with_stub_engine(tmp_path, backend, port):
    """bench.main's multi-rank control flow at the driver's largest size (8 ranks, launched exactly as the driver launches them:
    torch.distributed.run, one process per rank), on CPU ranks with a stub in place of the GPU engine: stream sharding (24 global
    streams, stream i -> rank i mod 8), barriers, the MAX / SUM reductions, the gather of detections, rank-0-only printing.
      rccl  the product's path: no torch in the ranks; the communicator is the product's host-side TCP star (comm.TcpComm: the
            control plane of open_comm and its fallback transport) in place of RCCL, which needs GPUs
      gloo  the torch.distributed rehearsal backend
      tcpfallback  as rccl, but the communicator reports what comm.open_comm reports after a failed RCCL bring-up: the line must say
            "dist_backend_ok": false and "rccl_ranks": 0 at its top level
    What stays unproven without an 8-GPU node is RCCL itself (one rank: tools/runs/r4_dist1.sh)."""
    script = tmp_path / "bench_worker.py"
    script.write_text(BENCH_WORKER % REPO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script), backend],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "BENCH8_OK" in out.stdout
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 8      # ONE JSON line, from rank 0


RDV_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
from async_ev_cnn_amd import comm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
made = []
uid = comm.exchange_id(rank, world, lambda: made.append(1) or bytes((7 * i + 3) %% 256 for i in range(comm.ID_BYTES)))
assert len(uid) == comm.ID_BYTES and uid == bytes((7 * i + 3) %% 256 for i in range(comm.ID_BYTES))
assert (len(made) == 1) == (rank == 0)          # only rank 0 creates the id
print("RDV_OK %%d" %% rank)
'''


@pytest.mark.parametrize("world,busy", [(2, False), (8, False), (3, True)])
def test_unique_id_rendezvous(tmp_path, world, busy):
    """The side channel of ncclUniqueId (async_ev_cnn_amd/comm.py: exchange_id): rank 0 creates the 128-byte id and serves it on
    MASTER_ADDR : MASTER_PORT + 1, the other ranks fetch it (retrying until rank 0 listens).  Plain processes, no launcher.
    busy: something else already listens on MASTER_PORT + 1 (it hangs up on whoever connects): rank 0 takes the next free port of
    the range and the others find it there."""
    import socket
    import threading
    script = tmp_path / "rdv_worker.py"
    script.write_text(RDV_WORKER % REPO)
    base = 29700 + world + (40 if busy else 0)
    squatter = None
    if busy:
        squatter = socket.socket()
        squatter.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        squatter.bind(("127.0.0.1", base + 1))
        squatter.listen(16)

        def hang_up():
            while True:
                try:
                    c, _ = squatter.accept()
                    c.close()
                except OSError:
                    return
        threading.Thread(target=hang_up, daemon=True).start()
    procs = []
    for r in reversed(range(world)):        # rank 0 LAST: the others must wait for it
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(base))
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert sorted(o.strip() for o in outs) == sorted("RDV_OK %d" % r for r in range(world))
    if squatter is not None:
        squatter.close()


OPEN_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np
from async_ev_cnn_amd import comm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
msgs = []
c, backend = comm.open_comm(0, rccl_timeout=120.0, log=msgs.append)
# no GPU in this container: no rank's probe process gets an RCCL communicator, every rank must end up on the star -- and say why
assert isinstance(c, comm.TcpComm) and backend.startswith("tcp: ") and len(backend) > 8 and msgs, (backend, msgs)
assert c.rccl_ranks == 0 and "NO RCCL COMMUNICATOR" in msgs[0]
assert c.rank == rank and c.world == world
assert float(c.allreduce([rank + 1.0], comm.SUM)[0]) == world * (world + 1) / 2
assert float(c.allreduce([rank + 1.0], comm.MAX)[0]) == world
c.barrier()
class Net:                      # stream i -> rank i mod world: rank r holds 2 streams, rank 0 one more
    n_streams = 3 if rank == 0 else 2
    def detect(self, *a):
        return np.full((self.n_streams, 70, 7), rank + 1.0, np.float32)
g, counts = c.gather_detections(Net(), 100, 2, 0.1, True, 0.0)
assert counts == [3] + [2] * (world - 1) and g.shape == (world, 3, 70, 7)
for r in range(world):
    assert np.all(g[r, :counts[r]] == r + 1.0) and np.all(g[r, counts[r]:] == 0.0)
assert c.bcast_bytes(b"from rank 0" if rank == 0 else b"ignored") == b"from rank 0"
c.close()
print("OPEN_OK %%d" %% rank)
'''


def test_open_comm_falls_back_to_the_tcp_star_on_every_rank(tmp_path):
    """comm.open_comm among three plain processes on a machine without a GPU: the TCP star comes up, rank 0's ncclUniqueId (or its
    failure) travels over it, no rank gets an RCCL communicator, the ranks agree on that and all of them return the star -- with
    the reason in the backend string bench.py prints as config.dist_backend.  Then the star's collectives: SUM / MAX all-reduce,
    barrier, the gather of ragged per-rank detections with zero padding, broadcast."""
    script = tmp_path / "open_worker.py"
    script.write_text(OPEN_WORKER % REPO)
    world, procs = 3, []
    for r in reversed(range(world)):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29810")
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    # (RCCL may print a line of its own when a process without a GPU exits)
    assert sorted(l for o in outs for l in o.splitlines() if l.startswith("OPEN_OK")) == sorted("OPEN_OK %d" % r for r in range(world)), outs


TWO_JOBS_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
from async_ev_cnn_amd import comm
rank, world, job = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["JOB"])
c = comm.TcpComm.from_env(timeout=60.0)
# every rank contributes its job number: a star that mixed the jobs would not sum to world * job
assert float(c.allreduce([float(job)], comm.SUM)[0]) == world * job, "ranks of two jobs met in one star"
uid = comm.exchange_id(rank, world, lambda: bytes([job]) * comm.ID_BYTES, timeout=60.0)
assert uid == bytes([job]) * comm.ID_BYTES, "the id of the other job"
c.barrier()
c.close()
print("JOB_OK %%d %%d" %% (job, rank))
'''


def test_two_jobs_with_adjacent_master_ports_do_not_mix(tmp_path):
    """Two jobs of the same world size on one node, MASTER_PORTs one apart: their rendezvous / control-plane port windows overlap almost
    entirely.  The second job's rank 0 starts late and the first job's rank 0 sits on the first port of the shared window, so job 2's
    clients probe job 1's rank 0 first -- the job token in the handshake turns them away (before it: same magic word, same world size
    -> accepted, and ranks of different jobs reduced and gathered together)."""
    import time
    script = tmp_path / "two_jobs_worker.py"
    script.write_text(TWO_JOBS_WORKER % REPO)
    world, procs = 3, []

    def start(job, port, ranks):
        for r in ranks:
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), JOB=str(job))
            procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    start(1, 29850, [0])            # job 1's rank 0 listens first: on the windows' first ports
    time.sleep(1.0)
    start(2, 29849, [1, 2])         # job 2's clients: their windows START one port below job 1's -- they meet job 1's rank 0 while probing
    start(1, 29850, [1, 2])
    time.sleep(1.0)
    start(2, 29849, [0])            # job 2's rank 0 comes last
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = sorted(l for o in outs for l in o.splitlines() if l.startswith("JOB_OK"))
    assert got == sorted("JOB_OK %d %d" % (j, r) for j in (1, 2) for r in range(world)), outs
