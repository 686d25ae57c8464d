"""Experiment: what does a queued run cost beyond its steps?  256 e-fcn streams in the steady state (after 560 steps), runs of K = 5 ... 80
steps timed back to back with a device sync in between: T(K) = a + b K.  (The driver's bench times ONE run of 20 steps.)
    python tools/exp_run_fixed_cost.py"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import async_ev_cnn_amd as A
import bench
from async_ev_cnn_amd import _native as NV

S, T0 = int(os.environ.get("AEV_FC_STREAMS", "256")), (60 if os.environ.get("AEV_FC_PLAN") else 560)
plan = [int(x) for x in os.environ["AEV_FC_PLAN"].split(",")] if os.environ.get("AEV_FC_PLAN") else [20, 5, 10, 20, 40, 80, 5, 10, 20, 40, 80, 20]
total = T0 + sum(plan)
with ThreadPoolExecutor(16) as pool:
    per = list(pool.map(lambda g: bench.make_stream("ncaltech", 1234 + g, total, 200), range(S)))
layers = A.layers_dict(A.net.EFCN_LAYERS)
net = A.EventNet(160, 224, layers, A.net.he_init_weights(layers, 1234), 'SAME', 0.1, 5e-5, S)
net.queue_upload([[per[s][t] for s in range(S)] for t in range(total)])
net.queue_run(0, T0, True); net.sync(); NV.device_sync(0)
t0, rows = T0, []
for K in plan:
    NV.device_sync(0)
    a = time.perf_counter()
    net.queue_run(t0, K, True); net.sync(); NV.device_sync(0)
    dt = time.perf_counter() - a
    rows.append((K, dt * 1e3))
    t0 += K
for K, ms in rows:
    print("K %3d: %.3f ms, %.4f ms per step" % (K, ms, ms / K))
Ks = np.array([r[0] for r in rows[1:]], float); Ts = np.array([r[1] for r in rows[1:]])
b, a = np.polyfit(Ks, Ts, 1)
print("fit: T(K) = %.3f ms + %.4f ms x K" % (a, b))
