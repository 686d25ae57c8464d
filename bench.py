#!/usr/bin/env python3
"""bench.py -- events/sec of the asynchronous e-fcn forward on N-Caltech101-shaped synthetic streams.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path -- the whole 13-layer async chain (integration -> 7 async conv + 5 async
max-pool), compute_all of the reference (src/models/event_numpy.py:75-79) -- over one batch of ~200 events
(configs/efcn_event.yml:8) for each of the S independent streams resident on the GPU.  Streams shard over ranks with
no data-path collective (weak scaling: S streams per GPU); the only collective is the final RCCL all-gather of the
predictions.  Events are resident in HBM before the timed region starts (device event queue).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel class, timed live with HIP events on the
library's own stream (aev_net_queue_profile); `cpu_baseline` is the CPU oracle (a plain-C port of the reference
algorithm, oracle/) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
F32_MFMA_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (= f32 vector peak)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=256, help="independent event streams per GPU")
    ap.add_argument("--batch", type=int, default=200, help="events per stream per step (efcn_event.yml:8)")
    ap.add_argument("--kind", default="ncaltech", choices=["ncaltech", "uniform", "dense1M"],
                    help="ncaltech: N-Caltech101-shaped saccade streams (config 2/4); uniform: uniform pixels 200 kev/s; "
                         "dense1M: uniform 1 Mevent/s stress (config 3)")
    ap.add_argument("--profile-steps", type=int, default=6)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the single-stream secondary measurement")
    ap.add_argument("--no-host-feed", action="store_true", help="skip the PCIe-inclusive secondary measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--conv-impl", default="auto", choices=["auto", "valu", "mfma"])
    ap.add_argument("--tuning", default="auto", choices=["auto", "throughput", "latency"],
                    help="kernel configuration policy (auto: by stream count)")
    ap.add_argument("--conv-math", default="f32", choices=["f32", "bf16x6"],
                    help="arithmetic of the tile-kernel conv layers (include/aev.h: aev_net_set_conv_math): f32 = the fmaf chain of the arithmetic "
                         "contract, bit-identical to the oracle (the headline); bf16x6 = exactly split operands, six bf16 products per product on "
                         "the bf16 MFMA -- as accurate against float64, not bit-identical")
    ap.add_argument("--dense", action="store_true",
                    help="BASELINE config 5 comparator: dense-frame mode (every site of every layer recomputed per step)")
    ap.add_argument("--dist-backend", default="rccl", choices=["rccl", "nccl", "gloo"],
                    help="rccl (= nccl) = the library's own RCCL communicator over xGMI (aev_comm_*, no PyTorch in the process: the "
                         "driver's multi-GPU runs); gloo = torch.distributed on CPU tensors, for rehearsing the N>1 control flow on a "
                         "box with fewer GPUs than ranks")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (rendezvous, communicator, reductions, gather) even with one rank: rehearses the "
                         "N > 1 path -- RCCL init, all-reduce, all-gather of detections -- on a one-GPU box")
    ap.add_argument("--sustained-steps", type=int, default=500,
                    help="steps of the `sustained` block (timed as ONE queued run between the cold-start measurement and the headline: "
                         "~1 s of device time, outside the clock ramp of an idle GPU); 0 skips it")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-frame-mode secondary measurement (BASELINE config 5 speed-up)")
    ap.add_argument("--sweep-batch", action="store_true",
                    help="instead of the bench line: events/s at B = 1, 20, 200, 1000, 10000 events per stream-step on the reference's "
                         "timing protocol (uniform events, tests/golden/ref_timing.json), one stream and --streams streams")
    ap.add_argument("--no-x6", action="store_true", help="skip the `split_bf16_conv` secondary measurement (the conv layers on split-bf16 arithmetic)")
    ap.add_argument("--no-legacy", action="store_true", help="skip the `legacy_protocol` block (rounds 1-3's measurement, for comparisons across rounds)")
    ap.add_argument("--legacy-protocol", action="store_true",
                    help="rounds 1-3's measurement, for comparisons across rounds: streams generated for W + K + P steps only, with the "
                         "saccade stretched over the stream (a 31-step stream moves ten times faster than N-Caltech101's), `value` "
                         "timed on steps W .. W + K of the idle GPU; no sustained block")
    return ap.parse_args(argv)


class Engine:
    """What bench.py needs from the product: the real one binds the HIP library; tests/test_dist_cpu.py drives the same control
    flow (sharding, barriers, MAX / SUM reductions, gather, rank-0-only printing) with a stub on CPU ranks."""

    def __init__(self):
        import async_ev_cnn_amd as A
        from async_ev_cnn_amd import _native as NV
        self.A, self.NV = A, NV

    def device_count(self):
        return self.NV.device_count()

    def device_sync(self, dev):
        self.NV.device_sync(dev)

    def runtime_info(self):
        return self.NV.runtime_info()

    def make_net(self, n_streams, dev, tuning="auto", conv_math="f32"):
        A = self.A
        layers = A.layers_dict(A.net.EFCN_LAYERS)
        return A.EventNet(160, 224, layers, A.net.he_init_weights(layers, 1234), 'SAME', 0.1, 5e-5, n_streams, device=dev, tuning=tuning, conv_math=conv_math)

    def probe_comm(self):
        """Before this rank touches the GPU: the TCP star of the job's ranks and the RCCL bring-up rehearsed in child processes
        (comm.probe_rccl) -> (star, "" or why RCCL cannot be used), handed to make_comm."""
        from async_ev_cnn_amd import comm
        tcp = comm.TcpComm.from_env()
        return tcp, comm.probe_rccl(tcp)

    def make_comm(self, dev, probed=None):
        """The communicator of this job's ranks: the library's RCCL communicator (aev_comm_create) if every rank gets one, else the
        host-side TCP star on every rank (comm.open_comm) -- the line then says "dist_backend_ok": false and "rccl_ranks": 0."""
        from async_ev_cnn_amd import comm
        c, backend = comm.open_comm(dev, log=lambda m: print(m, file=sys.stderr, flush=True), probed=probed)
        c.backend = backend
        return c

    def detections(self, net):
        """[S, 70, 7] detections of every stream (host array): what the gloo rehearsal gathers."""
        return net.detect(100, 2, 0.1, True, 0.0)


class NoDist:
    """One rank: no communicator, the "gather" is the local tensor."""
    backend = "none"
    rccl_ranks = 0      # ranks RCCL has seen (aev_comm_world of the communicator in use; 0: none in use)

    def __init__(self, eng, dev, probed=None):
        self.eng, self.dev, self.rank, self.world = eng, dev, 0, 1

    def barrier(self):
        pass

    def reduce(self, value, op):
        return float(value)

    def gather(self, net):
        d = self.eng.detections(net)
        return d[None], [int(d.shape[0])]

    def close(self):
        pass


class RcclDist(NoDist):
    """N ranks, one per GPU: the library's own RCCL communicator (include/aev.h: aev_comm_*) -- no PyTorch in the process, so every
    rank runs the library on the HIP runtime it was built for."""
    backend = "rccl"

    def __init__(self, eng, dev, probed=None):
        self.eng, self.dev = eng, dev
        self.comm = eng.make_comm(dev, probed) if probed is not None else eng.make_comm(dev)
        self.rank, self.world = self.comm.rank, self.comm.world
        self.backend = getattr(self.comm, "backend", "rccl")    # ("tcp: <why>" if RCCL could not be brought up on every rank)
        self.rccl_ranks = int(getattr(self.comm, "rccl_ranks", 0))

    def barrier(self):
        self.comm.barrier()

    def reduce(self, value, op):
        return float(self.comm.allreduce([float(value)], {"max": 1, "sum": 0}[op])[0])

    def gather(self, net):
        return self.comm.gather_detections(net, 100, 2, 0.1, True, 0.0)

    def close(self):
        self.comm.close()


class GlooDist(NoDist):
    """CPU rehearsal of the N > 1 control flow (torch.distributed over gloo, CPU tensors): the only place bench.py imports torch."""
    backend = "gloo"

    def __init__(self, eng, dev, probed=None):
        import torch
        import torch.distributed as dist
        self.eng, self.dev, self.torch, self.dist = eng, dev, torch, dist
        dist.init_process_group("gloo")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def reduce(self, value, op):
        t = self.torch.tensor([float(value)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather(self, net):
        from async_ev_cnn_amd import streams as S
        alld, counts = S.gather_predictions(self.torch.as_tensor(np.asarray(self.eng.detections(net))), force=True)
        return alld.numpy(), counts

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


def make_stream(kind, seed, n_steps, batch, leg_us=100000):
    from async_ev_cnn_amd import streams as S
    if kind == "ncaltech":
        # N-Caltech101's event rate (~100k events / 300 ms) and saccade (three 100 ms legs, walked again and again by a longer stream:
        # the statistics of a stream do not depend on how many steps are asked for); the crop drops 20-30 % of the events
        n_ev = int(n_steps * batch / 0.72) + 4 * batch
        while True:
            b = S.ncaltech_like_batches(seed, (160, 224), batch, n_events=n_ev, duration_us=3 * n_ev, leg_us=leg_us)
            if len(b) >= n_steps:
                return b[:n_steps]
            n_ev = int(n_ev * 1.15) + batch
    period = batch if kind == "dense1M" else 5 * batch
    return S.uniform_batches(seed, (160, 224), n_steps, batch, period)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, total_steps):
    """The CPU oracle (oracle/aev_oracle.c, a scalar port of the reference algorithm) on this host: one thread on one stream, then
    one thread per core, each on its own stream (independent streams are the path's own parallelism; ctypes releases the GIL
    inside the C step).  Bounded by --cpu-seconds each."""
    import threading
    from oracle import oracle as O
    from tests.util import EFCN, build_chain, he_weights
    n_steps = max(total_steps, int(args.cpu_seconds / 0.02))

    def prepare(seed):
        orc = build_chain(O, EFCN, he_weights(np.random.RandomState(1234), EFCN), 160, 224, 5e-5, 0.1, "SAME")
        sample = make_stream(args.kind, seed, n_steps, args.batch)
        O.net_step(orc[-1], sample[0])  # warm-up step
        return orc, sample

    def run(orc, sample, budget, out):
        n_ev, n_st, t0 = 0, 0, time.perf_counter()
        for ev in sample[1:]:
            O.net_step(orc[-1], ev)
            n_ev += len(ev)
            n_st += 1
            if time.perf_counter() - t0 > budget:
                break
        out.append((n_ev, n_st, time.perf_counter() - t0))

    orc, sample = prepare(1234)
    one = []
    run(orc, sample, args.cpu_seconds, one)
    n_ev, n_st, tc = one[0]
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))      # (a one-GPU box's CPU share is 16)
    prepared = [(orc, sample)] + [prepare(1234 + i) for i in range(1, cores)]
    res, threads = [], []
    ta = time.perf_counter()
    for o, smp in prepared:
        th = threading.Thread(target=run, args=(o, smp, args.cpu_seconds, res))
        th.start()
        threads.append(th)
    for th in threads:
        th.join()
    wall = time.perf_counter() - ta
    all_ev = sum(r[0] for r in res)
    # The port is NOT as fast as the reference it restates: a scalar fmaf chain in the arithmetic contract's order against the reference's
    # Cython ops + NumPy / OpenBLAS GEMMs.  tests/golden/ref_timing.json holds the ratio measured in the build container on identical
    # events (gen_golden.py --timing: port events/s : reference events/s); the reference itself cannot run on this host.
    ref_t = {}
    try:
        ref_t = json.load(open(os.path.join(REPO, "tests", "golden", "ref_timing.json")))
    except (OSError, ValueError):
        pass
    por = (ref_t.get("port_over_reference") or {}).get("ratio")
    ref_b200 = (ref_t.get("B200") or {}).get("events_per_s")
    return {"value": round(n_ev / tc, 1), "unit": "events/s", "cores": 1, "kind": "port",
            "kind_detail": "oracle/aev_oracle.c: a scalar float32 fmaf chain (bit-identical to the kernels), single thread -- SLOWER than the "
                           "reference's Cython + OpenBLAS path it restates: see port_over_reference",
            "port_over_reference": round(por, 4) if por else None,
            "reference_equivalent_events_per_s": round(n_ev / tc / por, 1) if por else None,
            "reference_in_build_container_events_per_s": round(ref_b200, 1) if ref_b200 else None,
            "sample": "one stream of the same workload (seed 1234), steps 1..%d (%d events) through oracle/aev_oracle.c, one thread, "
                      "%.1f s; reference_equivalent = value / port_over_reference (tests/golden/ref_timing.json: port and reference timed "
                      "on identical events in the build container; the reference cannot travel to this host)" % (n_st, n_ev, tc),
            "ms_per_step": round(1e3 * tc / max(n_st, 1), 2),
            "all_cores": {"value": round(all_ev / wall, 1), "unit": "events/s", "cores": cores, "cpu_model": cpu_model(),
                          "sample": "%d threads, one independent stream each (seeds 1234..%d), %d events in %.1f s" % (
                              cores, 1234 + cores - 1, all_ev, wall)}}


def sweep_batch(args, eng, dev):
    """events/s against the events per stream-step (B = 1, 20, 200, 1000, 10000) on the protocol the reference was timed
    with (tests/golden/ref_timing.json: uniform events, ts ~ 5 us per event): one stream alone and --streams streams together."""
    ref = {}
    try:
        ref = json.load(open(os.path.join(REPO, "tests", "golden", "ref_timing.json")))
    except OSError:
        pass
    rows = []
    for B in (1, 20, 200, 1000, 10000):
        for S_ in sorted(set([1, args.streams])):
            W, K = 3, (10 if B >= 10000 else 20)
            net = eng.make_net(S_, dev)
            per = [make_stream("uniform", 1234 + s, W + K, B) for s in range(S_)]
            net.queue_upload([[per[s][t] for s in range(S_)] for t in range(W + K)])
            net.queue_run(0, W, True)
            net.sync()
            t0 = time.perf_counter()
            net.queue_run(W, K, True)
            net.sync()
            dt = time.perf_counter() - t0
            rows.append({"events_per_stream_step": B, "streams": S_, "events_per_s": round(B * S_ * K / dt, 1), "ms_per_step": round(1e3 * dt / K, 4),
                         "reference_cpu_events_per_s": round(ref.get("B%d" % B, {}).get("events_per_s", 0), 1) or None})
            del net
    print(json.dumps({"metric": "events/sec on uniform synthetic streams (e-fcn) against the events per stream-step", "unit": "events/s",
                      "reference_host": ref.get("host"), "runtime": eng.runtime_info(), "sweep": rows}))


def main(argv=None, engine=None):
    args = parse(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # No PyTorch in the process, whatever N: the multi-rank collectives are the library's own RCCL communicator (aev_comm_*), so
    # every rank runs the library on the HIP runtime it was built for (/opt/rocm, ROCm 7.2).  torch is imported only by the gloo
    # rehearsal of the control flow on CPU ranks (--dist-backend gloo).  The line printed at the end says which runtime was mapped.
    multi = world > 1 or args.force_dist
    out_fd = None
    if multi:
        # RCCL prints a version banner on STDOUT when a communicator is created; the contract is ONE JSON line on stdout:
        # fd 1 points at stderr until the line is written
        sys.stdout.flush()
        out_fd = os.dup(1)
        os.dup2(2, 1)

    def restore_stdout():
        nonlocal out_fd
        if out_fd is not None:
            sys.stdout.flush()
            os.dup2(out_fd, 1)
            os.close(out_fd)
            out_fd = None

    eng = engine if engine is not None else Engine()
    from async_ev_cnn_amd import streams as S

    K, W, P = args.steps, args.warmup, (args.profile_steps if rank == 0 else 0)
    legacy = args.legacy_protocol
    KS = max(0, args.sustained_steps) if not (args.sweep_batch or legacy) else 0
    # queue layout: [W warm-up | K cold-start | KS sustained | W warm-up | K HEADLINE | P per-kernel pass]   (legacy: [W | K | P])
    total_steps = W + K + P if legacy else W + K + KS + W + K + P
    S_local = args.streams
    my_streams = S.shard_streams(S_local * world, rank, world)  # global stream ids of this rank
    if not args.sweep_batch:  # (synthetic streams: host work, before anything touches the GPU)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(len(os.sched_getaffinity(0)), 16))) as pool:
            per_stream = list(pool.map(lambda gid: make_stream(args.kind, 1234 + gid, total_steps, args.batch, None if legacy else 100000), my_streams))

    # N > 1 over RCCL: the ranks find each other and REHEARSE the RCCL bring-up in child processes before this process touches the GPU
    # (a bring-up that hangs is then a killed child, not a thread stuck inside RCCL beside the measurement)
    probed = eng.probe_comm() if multi and args.dist_backend != "gloo" and hasattr(eng, "probe_comm") else None
    n_dev = eng.device_count()
    if n_dev <= 0:
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    dev_index = local_rank % n_dev
    if args.sweep_batch:
        restore_stdout()
        return sweep_batch(args, eng, dev_index)
    dist = (GlooDist if args.dist_backend == "gloo" else RcclDist)(eng, dev_index, probed) if multi else NoDist(eng, dev_index)

    net = eng.make_net(S_local, dev_index, args.tuning, args.conv_math) if args.conv_math != "f32" else eng.make_net(S_local, dev_index, args.tuning)
    net.set_conv_impl(args.conv_impl)
    if args.dense:
        net.set_dense(True)
    steps = [[per_stream[s][t] for s in range(S_local)] for t in range(total_steps)]
    net.queue_upload(steps)
    graph = not args.no_graph

    def barrier():
        eng.device_sync(dev_index)       # hipDeviceSynchronize: every stream of the handle has drained
        dist.barrier()

    def timed_run(t0, n):
        """n queue steps from t0 as one queued run between barriers; (max-over-ranks seconds, events of all ranks, sites, events in)."""
        net.stats(reset=True)
        barrier()
        ta = time.perf_counter()
        net.queue_run(t0, n, graph)
        net.sync()
        eng.device_sync(dev_index)
        dt = time.perf_counter() - ta
        st, ev_in, _ = net.stats(reset=True)
        mine = sum(len(b) for t in range(t0, t0 + n) for b in steps[t])
        assert ev_in == mine, (ev_in, mine)
        dt_all, ev_all = dist.reduce(dt, "max"), dist.reduce(float(mine), "sum")
        barrier()
        return dt_all, ev_all, st

    t = 0
    net.queue_run(t, W, graph)      # warm-up (also captures the step graphs)
    net.sync()
    t += W
    # cold start: the K steps right behind the first warm-up, on a GPU that has been idle (what rounds 1-3 reported as `value`:
    # 5 + 20 steps = 50 ms lie inside the clock ramp of an idle MI355X)
    cold_dt, cold_ev, cold_sites = timed_run(t, K)
    t += K
    sustained = None
    if KS:
        sus_dt, sus_ev, _ = timed_run(t, KS)
        sustained = {"steps": KS, "ms_per_step": round(1e3 * sus_dt / KS, 4), "events_per_s": round(sus_ev / sus_dt, 1),
                     "seconds": round(sus_dt, 3), "note": "one queued run of %d steps (graph cascade), all ranks between barriers" % KS}
        t += KS
    if legacy:
        elapsed, total_events, sites = cold_dt, cold_ev, cold_sites
    else:
        net.queue_run(t, W, graph)      # the W untimed warm-up steps of the headline
        net.sync()
        t += W
        elapsed, total_events, sites = timed_run(t, K)          # the K timed steps: `value`
        t += K
    timed_events = sum(len(b) for tt in range(t - K, t) for b in steps[tt])
    out_headline = net.output() if (world == 1 and args.conv_math == "f32" and not args.no_x6 and engine is None) else None  # [S,5,7,110] after step t
    cold_start = {"ms_per_step": round(1e3 * cold_dt / K, 4), "events_per_s": round(cold_ev / cold_dt, 1), "steps": K,
                  "note": "the same streams' steps %d..%d on the idle GPU: the surfaces are still filling up (fewer active sites per step)" % (W, W + K - 1)}

    # the one collective of the path (outside the timed region, reported separately): every stream's output is decoded
    # on the GPU (YOLO decode + NMS, the runner's thresholds runner.py:36-39) and the detections [S,70,7] are
    # all-gathered -- RCCL over xGMI for N > 1 (BASELINE config 4: "RCCL gather of detections")
    c, h, w = net.shapes[-1]
    tg0 = time.perf_counter()
    alld, counts = dist.gather(net)
    eng.device_sync(dev_index)
    gather_ms = 1e3 * (time.perf_counter() - tg0)
    ok = tuple(alld.shape) == (dist.world, S_local, h * w * 2, 7) and list(counts) == [S_local] * dist.world and bool(np.isfinite(alld).all())
    assert ok, "detections: bad shape or non-finite values"

    if rank != 0:
        dist.barrier()
        dist.close()
        restore_stdout()
        return None

    # ---- rank 0: per-kernel timing (live HIP events on the library's stream), roofline, CPU baseline -------------
    kernels = {}
    roof = None
    per_layer = None
    if P > 0:
        net.stats(reset=True)
        prof, prof_layers = net.queue_profile(t, P)
        psites, pev, _ = net.stats(reset=True)
        stream_steps = P * S_local
        a_bytes, a_flops, dense_bytes = net.algorithmic_bytes_flops(psites, pev, stream_steps)
        # algorithmic work per kernel class: every conv layer is booked on the class that actually ran it
        work = {}   # class -> [bound, amount]
        def book(cls, bound, amount):
            w = work.setdefault(cls, [bound, 0.0])
            w[1] += amount
        for i, kind in enumerate(net.kinds):
            cc, hh, ww = net.shapes[i]
            if kind == 'conv':
                kh, kw, ci, co = net.conv_specs[i]
                ran = [k for (ln, k) in prof_layers if ln == net.names[i] and k.startswith("k_conv_") and k != "k_conv_expand"]
                cls = ran[0] if ran else "k_conv_patch"
                if ci * kh * kw < 32:   # conv1 (K = 9): pure gather/scatter, no GEMM to speak of -> HBM-side accounting
                    book(cls, "hbm", float(psites[i]) * (8 * ci * kh * kw + 8 * co))
                else:
                    book(cls, "mfma", float(psites[i]) * 2 * (2 * co * ci * kh * kw))
            elif kind == 'pool':
                kh, kw = net.conv_specs[i][:2]
                book("k_pool_eval", "hbm", float(psites[i]) * cc * (8 * kh * kw + 4))
        # k_leak_sweep skips sites whose decay is exactly 0 (it reads their "may decay" flag byte only): measure that share on a sample of
        # streams so that the kernel is not credited with bytes it does not have to move
        skip_elems = tot_elems = 0
        for i, kind in enumerate(net.kinds):
            if kind == 'conv':
                for sidx in range(0, S_local, max(1, S_local // 8)):
                    a = net.read(i, 4, sidx)  # AEV_RAW_ACT [C,H,W]
                    z = (a.reshape(a.shape[0], -1) == 0).all(axis=0)     # whole sites (all channels) with zero decay
                    skip_elems += int(z.sum()) * a.shape[0]
                    tot_elems += a.size
        zero_share = skip_elems / max(tot_elems, 1)
        sweep_alg = (dense_bytes - 8 * 160 * 224) * stream_steps
        # (a site whose decay is zero costs the sweep one flag byte, not its 12 B per element)
        book("k_leak_sweep", "hbm", sweep_alg * (1.0 - zero_share))
        book("k_integ_sweep", "hbm", 8.0 * 160 * 224 * stream_steps)
        # k_post (booked as k_compact): one block per stream between two heavy kernels.  Its algorithmic bytes are the byte maps it reads
        # and clears, the ordered lists it writes and the marks it sets: per launch of layer i (input layer or conv)
        #   maps        2 B x H W per map (read + clear; conv: touch + sign-change map, input layer: event map)
        #   event list  4 B x events of the layer
        #   pool behind it: pending map 2 B x windows of the map, window list 4 B x listed windows, first-touch keys 8 B x listed windows
        #   consumer    kh kw B of touch marks per event that reaches it (+ 36 B per active tile: not counted)
        # A latency-bound kernel (dependent round trips inside one block per stream): the HBM fraction says how far from streaming it is.
        post_bytes = 0.0
        for i, kind in enumerate(net.kinds):
            if kind == 'pool':
                continue
            cc, hh, ww = net.shapes[i]
            b = (2.0 if kind == 'intgr' else 4.0) * hh * ww * stream_steps + 4.0 * float(psites[i])
            feed, j = float(psites[i]), i + 1
            if j < len(net.kinds) and net.kinds[j] == 'pool':
                pc, ph, pw = net.shapes[j]
                b += 2.0 * ph * pw * stream_steps + 12.0 * float(psites[j])
                feed, j = float(psites[j]), j + 1
            if j < len(net.kinds) and net.kinds[j] == 'conv':
                b += feed * net.conv_specs[j][0] * net.conv_specs[j][1]
            post_bytes += b
        book("k_compact", "hbm", post_bytes)
        attributed = {k: (v[0], v[1]) for k, v in work.items()}
        tot_ms = sum(v[0] for v in prof.values())
        for name, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
            ent = {"ms_total": round(ms, 4), "launches": cnt, "us_per_launch": round(1e3 * ms / cnt, 3),
                   "share": round(ms / tot_ms, 4)}
            if name in attributed and attributed[name][1]:
                bound, work = attributed[name]
                rate = work / (ms * 1e-3)
                if bound == "hbm":
                    ent.update(bound="hbm", achieved=round(rate / 1e9, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=round(rate / 1e9 / HBM_PEAK_GBS, 4))
                else:
                    ent.update(bound="mfma", achieved=round(rate / 1e12, 3), peak=F32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                               frac=round(rate / 1e12 / F32_MFMA_PEAK_TFLOPS, 4))
            kernels[name] = ent
        if "k_leak_sweep" in kernels:
            kernels["k_leak_sweep"]["zero_decay_share_skipped"] = round(zero_share, 4)
        if "k_compact" in kernels:
            kernels["k_compact"]["note"] = "k_post: one block per stream, latency-bound; bytes = maps read + cleared, lists written, marks set"
        per_layer = {}
        for (lname, kname), (ms, cnt) in prof_layers.items():
            per_layer.setdefault(lname, {})[kname] = round(1e3 * ms / cnt, 2)   # us per launch
        # the conv class layer by layer: flops of the layer's recomputed sites over its own launches
        conv_layers = {}
        for i, kind in enumerate(net.kinds):
            if kind != 'conv':
                continue
            kh, kw, ci, co = net.conv_specs[i]
            for (lname, kname), (ms, cnt) in prof_layers.items():
                if lname == net.names[i] and kname.startswith("k_conv_") and kname != "k_conv_expand" and ci * kh * kw >= 32 and ms > 0:
                    fl = float(psites[i]) * 2 * (2 * co * ci * kh * kw)
                    conv_layers[lname] = {"kernel": kname, "us_per_launch": round(1e3 * ms / cnt, 2), "gflop_per_launch": round(fl / cnt / 1e9, 3),
                                          "tflops": round(fl / (ms * 1e-3) / 1e12, 2), "frac": round(fl / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
        dom = next(iter(kernels))
        for name, ent in kernels.items():       # dominant = largest share among the kernels that carry a roofline
            if "bound" in ent:
                dom = name
                break
        d = kernels[dom]
        traffic = l2_hit = traffic_source = None
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):  # HBM bytes per launch from the rocprofv3 --pmc passes (tools/pmc_summary.py), same workload
            tj = json.load(open(pmc))
            if tj.get("streams_per_gpu") == S_local and tj.get("stream_kind") == args.kind:
                rows = [v for k, v in tj["kernels"].items() if k.startswith(dom)]
                if rows:
                    nl = sum(v["launches"] for v in rows)
                    traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rows) / nl)
                    if all("l2_hit_rate" in v for v in rows):
                        l2_hit = round(sum(v["l2_hit_rate"] * v["launches"] for v in rows) / nl, 4)
                    traffic_source = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum passes of this "
                                      "workload (%s), NOT collected in this run" % tj.get("collected", "see profiles/README.md"))
        roof = {"kernel": dom, "bound": d.get("bound"), "achieved": d.get("achieved"), "peak": d.get("peak"),
                "unit": d.get("unit"), "frac": d.get("frac"), "traffic": traffic, "l2_hit_rate": l2_hit, "traffic_source": traffic_source,
                "us_per_launch": d["us_per_launch"], "share_of_step": d["share"], "by_layer": conv_layers,
                "whole_step": {"algorithmic_GBps": round(a_bytes / (tot_ms * 1e-3) / 1e9, 2),
                               "frac_of_hbm_peak": round(a_bytes / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "algorithmic_TFLOPs": round(a_flops / (tot_ms * 1e-3) / 1e12, 3),
                               "bytes_per_stream_step": round(a_bytes / stream_steps), "flops_per_stream_step": round(a_flops / stream_steps)}}

    # ---- secondary figure: ONE stream alone on the GPU (the literal reading of BASELINE configs[1]; latency-bound) ----
    single = None
    if world == 1 and not args.no_single:
        one = eng.make_net(1, dev_index)
        KS1 = 400 if KS > 0 else 0      # (a run long enough for the pipeline's fill and drain -- ~0.19 ms per run -- not to show)
        sb = make_stream(args.kind, 1234, W + K + KS1, args.batch)
        one.queue_upload([[b] for b in sb])
        one.queue_run(0, W, graph)
        one.sync()
        ts0 = time.perf_counter()
        one.queue_run(W, K, graph)
        one.sync()
        ts1 = time.perf_counter()
        n1 = sum(len(b) for b in sb[W:W + K])
        single = {"events_per_s": round(n1 / (ts1 - ts0), 1), "ms_per_step": round(1e3 * (ts1 - ts0) / K, 4), "steps": K}
        if KS1:
            tq0 = time.perf_counter()
            one.queue_run(W + K, KS1, graph)
            one.sync()
            tq1 = time.perf_counter()
            single["sustained"] = {"steps": KS1, "ms_per_step": round(1e3 * (tq1 - tq0) / KS1, 4),
                                   "events_per_s": round(sum(len(b) for b in sb[W + K:]) / (tq1 - tq0), 1),
                                   "note": "one queued run: a step's two halves are pipeline stages, a run costs ~0.19 ms of fill and drain beyond its steps"}
        del one
        # per-event mode (the reference's actual promise: asynchronous per-event processing, runner.py:65-72 batches for convenience): one
        # event per step on one stream, uniform pixels ~5 us apart (the protocol of --sweep-batch / tests/golden/ref_timing.json)
        pe = eng.make_net(1, dev_index)
        pb = make_stream("uniform", 1234, 8 + 64, 1)
        pe.queue_upload([[b] for b in pb])
        pe.queue_run(0, 8, graph)
        pe.sync()
        tp0 = time.perf_counter()
        pe.queue_run(8, 64, graph)
        pe.sync()
        single["per_event_us"] = round(1e6 * (time.perf_counter() - tp0) / 64, 2)
        single["per_event_note"] = "64 steps of ONE event each in one queued run (uniform pixels): microseconds per event = per step"
        del pe

    # ---- secondary figure: the same steps fed from HOST buffers through aev_net_step (PCIe upload inside the timing) ----
    host_feed = None
    if world == 1 and not args.no_host_feed:
        net.reset()
        for tt in range(W):
            net.step(steps[tt])
        net.sync()
        th0 = time.perf_counter()
        for tt in range(W, W + K):
            net.step(steps[tt])
        net.sync()
        th1 = time.perf_counter()
        hf_events = sum(len(b) for tt in range(W, W + K) for b in steps[tt])
        host_feed = {"events_per_s": round(hf_events / (th1 - th0), 1), "ms_per_step": round(1e3 * (th1 - th0) / K, 4),
                     "over_cold_start_ms_per_step": round((th1 - th0) / cold_dt, 3),
                     "note": "events handed over as host arrays every step (aev_net_step): NumPy concatenation + pinned staging + "
                             "H2D copy + eager launches inside the timed region; timed on steps %d..%d from a reset state -- the steps of "
                             "`cold_start` (surfaces still filling), NOT the headline's: compare with cold_start.ms_per_step" % (W, W + K - 1)}

    # ---- secondary figure: BASELINE config 5 -- the same steps in dense-frame mode (every site of every layer recomputed) ----
    dense_frame = None
    if world == 1 and not args.no_dense and not args.dense:
        dn = eng.make_net(S_local, dev_index, args.tuning)
        dn.set_dense(1)
        dn.queue_upload(steps[:t])             # (the same steps up to the end of the headline's K: the same surface in every step)
        dn.queue_run(0, t - K, graph)
        dn.sync()
        td0 = time.perf_counter()
        dn.queue_run(t - K, K, graph)
        dn.sync()
        td = time.perf_counter() - td0
        dense_vs = None
        if out_headline is not None:    # what float32 rounding alone does to the output: the asynchronous path against the dense recompute of the same surface
            dd = np.abs(dn.output().astype(np.float64) - out_headline.astype(np.float64))
            dense_vs = {"max_abs_diff": float(dd.max()), "share_within_1e-5": round(float((dd <= 1e-5 + 1e-5 * np.abs(out_headline)).mean()), 6)}
        dense_frame = {"ms_per_step": round(1e3 * td / K, 4), "events_per_s": round(timed_events / td, 1), "output_vs_headline": dense_vs,
                       "async_speedup_over_dense_frame": round(td / elapsed, 3),
                       "note": "aev_net_set_dense(1): no leak extrapolation, every site of every layer recomputed from the surface each step, "
                               "same kernels, same streams and events (frame_numpy.py:62-88 is the reference's dense model)"}
        del dn

    # ---- secondary figure: the same steps with the tile-kernel conv layers on SPLIT-bf16 arithmetic (aev_net_set_conv_math(AEV_MATH_BF16X6)) ----
    # NOT the headline: `value` is measured on the f32 fmaf-chain path, bit-identical to the oracle.  This mode splits every f32 operand exactly
    # into three bf16 parts and accumulates six bf16 products per product in f32 on the bf16 MFMA -- as accurate against float64 as the chain
    # (tools/probes/mfma_bf16x6.hip), within north_star's 1e-5 of the oracle (tests/test_gpu_x6.py), bit-exact on the exact fixture family.
    x6_block = None
    if out_headline is not None:
        xn = eng.make_net(S_local, dev_index, args.tuning, "bf16x6")
        xn.queue_upload(steps[:t])
        xn.queue_run(0, t - K, graph)
        xn.sync()
        eng.device_sync(dev_index)
        tx0 = time.perf_counter()
        xn.queue_run(t - K, K, graph)
        xn.sync()
        eng.device_sync(dev_index)
        tx = time.perf_counter() - tx0
        out_x6 = xn.output()
        diff = np.abs(out_x6.astype(np.float64) - out_headline.astype(np.float64))
        scale = float(np.abs(out_headline).max())
        x6_block = {"ms_per_step": round(1e3 * tx / K, 4), "events_per_s": round(timed_events / tx, 1), "speedup_over_headline": round(elapsed / tx, 3),
                    "conv_math": "bf16x6: conv2 ... conv5 of the e-fcn on v_mfma_f32_16x16x32_bf16 with exactly split operands (csrc/aev_conv_x6.hpp)",
                    "output_vs_headline": {"max_abs_diff": float(diff.max()), "max_abs_output": scale, "max_abs_diff_over_max_abs_output": float(diff.max() / max(scale, 1e-30)),
                                           "share_within_1e-5": round(float((diff <= 1e-5 + 1e-5 * np.abs(out_headline)).mean()), 6),
                                           "note": "the [S,5,7,110] outputs of both arithmetics after the same %d steps of the same streams; the yardstick for "
                                                   "what float32 rounding alone does over that many asynchronous steps is dense_frame.output_vs_headline "
                                                   "(the f32 asynchronous path against the f32 dense recompute of the same surface)" % t},
                    "note": "NOT `value`: the headline is the f32 fmaf-chain path (bit-identical to the oracle); this mode is within 1e-5 of it on the tests' sampled streams (all outputs of 256 streams after 6 steps: 99.9999 % within, max 1.6e-5) "
                            "(tests/test_gpu_x6.py), bit-exact on the exact fixture family"}
        del xn

    # ---- secondary figure: rounds 1-3's protocol on this build (continuity across rounds; see --legacy-protocol) ----
    legacy_block = None
    if world == 1 and not args.no_legacy and not legacy and not args.dense:
        ln = eng.make_net(S_local, dev_index, args.tuning)
        ln.set_conv_impl(args.conv_impl)
        lsteps_n = W + K
        with ThreadPoolExecutor(max(1, min(len(os.sched_getaffinity(0)), 16))) as pool:
            lper = list(pool.map(lambda gid: make_stream(args.kind, 1234 + gid, lsteps_n + 6, args.batch, None), my_streams))
        lsteps = [[lper[s][tt] for s in range(S_local)] for tt in range(lsteps_n)]
        ln.queue_upload(lsteps)
        ln.queue_run(0, W, graph)
        ln.sync()
        ln.stats(reset=True)
        eng.device_sync(dev_index)
        tl0 = time.perf_counter()
        ln.queue_run(W, K, graph)
        ln.sync()
        eng.device_sync(dev_index)
        tl = time.perf_counter() - tl0
        lsites, lev, _ = ln.stats(reset=True)
        legacy_block = {"ms_per_step": round(1e3 * tl / K, 4), "events_per_s": round(lev / tl, 1), "steps": K,
                        "mean_active_sites_per_stream_step": {net.names[i]: round(float(lsites[i]) / (K * S_local), 1) for i in (1, 2, 3, 5) if i < len(net.names)},
                        "note": "rounds 1-3's measurement on this build: %d-step streams whose saccade is stretched over the stream (ten times "
                                "N-Caltech101's speed), steps %d..%d from a reset state on an idle GPU" % (lsteps_n + 6, W, W + K - 1)}
        del ln

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, W + K)

    value = total_events / elapsed
    mean_sites = {net.names[i]: round(float(sites[i]) / (K * S_local), 1) for i in range(len(net.names))}
    line = {
        "metric": "events/sec on N-Caltech101-shape stream (e-fcn), 1->8 GPU; achieved HBM GB/s",
        "value": round(value, 1), "unit": "events/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(1e3 * elapsed / K, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # N > 1: did RCCL see all the ranks?  (false: the ranks fell back to the host-side TCP star for barriers, reductions and the final
        # gather -- config.dist_backend says why; the streams themselves never communicate, so `value` is still N GPUs' throughput)
        "rccl_ranks": dist.rccl_ranks, "dist_backend_ok": (not multi) or dist.backend in ("rccl", "gloo"),
        "config": {"workload": "e-fcn (configs/efcn_event.yml) async forward, 160x224 frame (180x240 sensor center-cropped), "
                               "%s synthetic streams, %d events/stream/step, random-init weights (rule W, seed 1234), "
                               "%d independent streams per GPU advanced together" % (args.kind, args.batch, S_local),
                   "streams_per_gpu": S_local, "events_per_stream_step": args.batch, "stream_kind": args.kind,
                   "parallelism": "streams sharded over %d GPU(s), no data-path collective" % world,
                   "hipgraph": graph, "conv_impl": args.conv_impl, "conv_math": args.conv_math, "dense_frame_mode": bool(args.dense),
                   "mean_active_sites_per_stream_step": mean_sites,
                   "detect_and_gather_ms": round(gather_ms, 3), "dist_backend": dist.backend,
                   "timed_region": ("legacy protocol (rounds 1-3): steps %d..%d of %d-step streams, idle GPU" % (W, W + K - 1, total_steps)) if legacy else
                                   ("steps %d..%d of the streams: %d warm-up + %d cold-start + %d sustained steps, then the %d untimed warm-up "
                                    "steps and the %d timed steps of `value`" % (t - K, t - 1, W, K, KS, W, K))},
        "runtime": eng.runtime_info(),
        "roofline": roof, "cpu_baseline": cpu, "sustained": sustained, "cold_start": cold_start, "legacy_protocol": legacy_block,
        "single_stream": single, "host_fed": host_feed, "dense_frame": dense_frame, "split_bf16_conv": x6_block, "kernels": kernels,
        "us_per_launch_by_layer": per_layer,
    }
    if cpu:
        line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
        line["speedup_vs_cpu_all_cores"] = round(value / cpu["all_cores"]["value"], 1)
        if cpu.get("reference_equivalent_events_per_s"):     # (the figure to quote: against what the reference would do on this host)
            line["speedup_vs_reference_equivalent"] = round(value / cpu["reference_equivalent_events_per_s"], 1)
        if single:
            single["speedup_vs_cpu_baseline"] = round(single["events_per_s"] / cpu["value"], 1)
            if cpu.get("reference_equivalent_events_per_s"):
                single["speedup_vs_reference_equivalent"] = round(single["events_per_s"] / cpu["reference_equivalent_events_per_s"], 1)
    dist.barrier()
    dist.close()
    restore_stdout()
    print(json.dumps(line), flush=True)
    return line


if __name__ == "__main__":
    main()
