"""Experiment (round 4): what do two HALF batches side by side give?  256 streams as G handles of 256 / G streams,
every handle's queued run submitted by a host thread of its own (graph replays enqueued by ONE host thread do not overlap on this
runtime, docs/KERNEL_NOTES.md round 3), against one handle of 256.  The handles are uncoordinated: whatever overlap there is comes
from the hardware scheduler placing one group's bookkeeping / bandwidth-bound kernels beside the other group's tile kernels.
  python tools/exp_half_batches.py [streams] [steps]
Plans per group: the default for its size (128 streams: few-streams plan) and the many-streams plan forced (AEV_OVERLAP=1
AEV_LOOKAHEAD=1); an optional start offset staggers the groups by part of a step."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import async_ev_cnn_amd as A
import bench

S_TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 80
W = 16
layers = A.layers_dict(A.net.EFCN_LAYERS)
w = A.net.he_init_weights(layers, 1234)
per_stream = [bench.make_stream("ncaltech", 1234 + g, W + K, 200) for g in range(S_TOTAL)]
ev = sum(len(per_stream[s][t]) for s in range(S_TOTAL) for t in range(W, W + K))


def run(G, env, stagger_ms=0.0, threads=True):
    for k, v in env.items():
        os.environ[k] = v
    Sg = S_TOTAL // G
    nets = []
    for g in range(G):
        net = A.EventNet(160, 224, layers, w, 'SAME', 0.1, 5e-5, Sg)
        net.queue_upload([[per_stream[g * Sg + s][t] for s in range(Sg)] for t in range(W + K)])
        net.queue_run(0, W, True)
        net.sync()
        nets.append(net)
    for k in env:
        del os.environ[k]

    def work(i, net):
        if stagger_ms and i:
            time.sleep(1e-3 * stagger_ms * i)
        net.queue_run(W, K, True)
        net.sync()
    t0 = time.perf_counter()
    if threads and G > 1:
        th = [threading.Thread(target=work, args=(i, net)) for i, net in enumerate(nets)]
        for t in th: t.start()
        for t in th: t.join()
    else:
        for net in nets: net.queue_run(W, K, True)
        for net in nets: net.sync()
    dt = time.perf_counter() - t0
    outs = [net.output() for net in nets]
    del nets
    return dt, outs


if os.environ.get("AEV_EXP_TRACE"):   # under rocprofv3 --kernel-trace: one configuration, few steps (tools/two_handles_timeline.py reads the trace)
    G = int(os.environ["AEV_EXP_TRACE"])
    dt, _ = run(G, {})
    print("%d x %d streams: %.3f ms per step of all streams (under the profiler)" % (G, S_TOTAL // G, 1e3 * dt / K))
    sys.exit(0)
base, ref = run(1, {})
print("1 x %d streams: %.3f ms per step, %.2f M events/s" % (S_TOTAL, 1e3 * base / K, ev / base / 1e6), flush=True)
import numpy as np
ref = ref[0]
for G in (2, 4):
    for name, env in (("default plan", {}), ("many-streams plan", {"AEV_OVERLAP": "1", "AEV_LOOKAHEAD": "1"})):
        for stg in (0.0, 0.5):
            dt, outs = run(G, env, stg)
            same = np.array_equal(np.concatenate(outs, 0), ref)
            print("%d x %d streams, a host thread each, %s, stagger %.1f ms: %.3f ms per step of all streams, %.2f M events/s (x%.3f), outputs identical: %s"
                  % (G, S_TOTAL // G, name, stg, 1e3 * dt / K, ev / dt / 1e6, base / dt, same), flush=True)
dt, _ = run(2, {}, 0.0, threads=False)
print("2 x %d streams, ONE host thread: %.3f ms per step" % (S_TOTAL // 2, 1e3 * dt / K), flush=True)
