#!/usr/bin/env python3
"""Work per layer of the bench workload: recomputed conv sites / evaluated pool windows per stream-step (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import async_ev_cnn_amd as A

S, W, K = 16, 6, 12
layers = A.layers_dict(A.net.EFCN_LAYERS)
net = A.EventNet(160, 224, layers, A.net.he_init_weights(layers, 1234), 'SAME', 0.1, 5e-5, S)
per = [bench.make_stream("ncaltech", 1234 + s, W + K, 200) for s in range(S)]
net.queue_upload([[per[s][t] for s in range(S)] for t in range(W + K)])
net.queue_run(0, W)
net.stats(reset=True)
net.queue_run(W, K)
sites, ev, st = net.stats()
print("events per stream-step %.1f" % (ev / (S * K)))
for name, n in zip(net.names, sites):
    print("%-8s %9.1f per stream-step" % (name, n / (S * K)))
