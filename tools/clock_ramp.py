#!/usr/bin/env python3
"""How long does the device take to reach its steady step time after an idle gap?  (ms per replayed step, in groups of 5.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import async_ev_cnn_amd as A

S, T = 256, 165
layers = A.layers_dict(A.net.EFCN_LAYERS)
net = A.EventNet(160, 224, layers, A.net.he_init_weights(layers, 1234), 'SAME', 0.1, 5e-5, S)
per = [bench.make_stream("ncaltech", 1234 + s, T, 200) for s in range(S)]
net.queue_upload([[per[s][t] for s in range(S)] for t in range(T)])
net.queue_run(0, 5)          # capture
net.sync()
time.sleep(3.0)              # idle gap (stream generation takes seconds in bench.py)
out = []
for t in range(5, T, 5):
    out.append(round(net.queue_run_timed(t, 5) / 5, 4))
print("ms per step, groups of 5 steps after a 3 s idle gap:", out)
