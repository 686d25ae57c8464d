"""How many of the windows a pool layer evaluates per step are windows of incoming events, how many are only PENDING (their winner is not
their slowest decayer: maxpool.py:140-142)?  e-fcn, N-Caltech101-like streams, 4 streams, after 300 steps.  (round 5: would evaluating the
pending windows ahead of the step, with the look-ahead sweep, take the pool evaluations off the critical chain?)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import async_ev_cnn_amd as A
from async_ev_cnn_amd import _native as N
import bench

S, T = 4, 300
layers = A.layers_dict(A.net.EFCN_LAYERS)
net = A.EventNet(160, 224, layers, A.net.he_init_weights(layers, 1234), 'SAME', 0.1, 5e-5, S)
per = [bench.make_stream("ncaltech", 1234 + s, T, 200) for s in range(S)]
net.queue_upload([[per[s][t] for s in range(S)] for t in range(T)])
net.queue_run(0, T - 1, use_graph=False)
pend_before = {li: [int(net.read(li, N.RECOMPUTE, s).astype(bool).sum()) for s in range(S)] for li, k in enumerate(net.kinds) if k == "pool"}
net.queue_run(T - 1, 1, use_graph=False)
for li, k in enumerate(net.kinds):
    if k != "pool":
        continue
    rows = []
    for s in range(S):
        y, x = net.layer_events(li, s)            # the windows the layer evaluated in the last step (incoming first, pending behind)
        cy, cx = net.layer_events(li - 1, s)      # the events of the conv layer below
        inc = len(set(zip((np.asarray(cy) // 2).tolist(), (np.asarray(cx) // 2).tolist())))
        shape = net.read(li, N.RECOMPUTE, s).shape
        rows.append((len(y), inc, pend_before[li][s], shape[0] * shape[1]))
    r = np.array(rows, dtype=float).mean(0)
    print("layer %2d (pool): windows evaluated %.0f of %d, of incoming events %.0f, pending before the step %.0f" % (li, r[0], r[3], r[1], r[2]))
