"""Experiment: does running the streams as G independent handles (own HIP streams) overlap one group's HBM-bound sweep
with another group's MFMA-bound conv kernels?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import async_ev_cnn_amd as A
import bench

S_TOTAL, K, W = 256, 40, 8
layers = A.layers_dict(A.net.EFCN_LAYERS)
w = A.net.he_init_weights(layers, 1234)
per_stream = [bench.make_stream("ncaltech", 1234 + g, W + K, 200) for g in range(S_TOTAL)]
for G in (1, 2, 4):
    Sg = S_TOTAL // G
    nets = []
    for g in range(G):
        net = A.EventNet(160, 224, layers, w, 'SAME', 0.1, 5e-5, Sg)
        steps = [[per_stream[g * Sg + s][t] for s in range(Sg)] for t in range(W + K)]
        net.queue_upload(steps)
        nets.append(net)
    for net in nets:
        net.queue_run(0, W, True)
    for net in nets:
        net.sync()
    ev = sum(len(per_stream[s][t]) for s in range(S_TOTAL) for t in range(W, W + K))
    t0 = time.perf_counter()
    # every group runs its K steps as ONE queued run (graph cascade, look-ahead, tail overlap inside each group); the calls are
    # asynchronous, so the groups' graphs execute side by side on their own HIP streams
    for net in nets:
        net.queue_run(W, K, True)
    for net in nets:
        net.sync()
    dt = time.perf_counter() - t0
    print("groups=%d: %.3f ms per step of all %d streams, %.2f M events/s" % (G, 1e3 * dt / K, S_TOTAL, ev / dt / 1e6))
    del nets
