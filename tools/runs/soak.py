"""Soak at bench scale (dev helper, GPU): 256 e-fcn streams x 43 steps through the production path (hipGraph cascade, look-ahead,
tail overlap) against the plain list-driven VALU kernels without look-ahead; every layer of a sample of streams, bit for bit.
    python tools/runs/soak.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import async_ev_cnn_amd as A  # noqa: E402
import bench  # noqa: E402
from async_ev_cnn_amd import _native as N  # noqa: E402

S, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), 43   # (e.g. 160, 8, 1: the few-streams plan)
per = [bench.make_stream("ncaltech", 777 + s, T, 200) for s in range(S)]
steps = [[per[s][t] for s in range(S)] for t in range(T)]
got = {}
for impl in ("auto", "valu"):
    if S >= 208: os.environ["AEV_LOOKAHEAD"] = "1" if impl == "auto" else "0"
    layers = A.layers_dict(A.net.EFCN_LAYERS)
    net = A.EventNet(160, 224, layers, A.net.he_init_weights(layers, 1234), 'SAME', 0.1, 5e-5, S)
    net.set_conv_impl(impl)
    net.queue_upload(steps)
    net.queue_run(0, 3, use_graph=True)          # 2 + 1
    net.queue_run(3, 29, use_graph=True)         # 8 + 8 + 8 + 4 + 1
    net.queue_run(32, T - 32, use_graph=True)    # 8 + 2 + 1
    got[impl] = [net.output()] + [net.read(li, kind, s) for s in sorted({0, (100 * S) // 256, S - 1}) for li in range(1, 13) for kind in (N.FEATUREMAP, N.CONV_ACTFN)]
    del net
bad = sum(0 if np.array_equal(x, y) else 1 for x, y in zip(got["auto"], got["valu"]))
print("soak: %d arrays compared, %d differ" % (len(got["auto"]), bad))
sys.exit(1 if bad else 0)
