#!/usr/bin/env python3
"""Debug helper: one seed of test_random_architectures_many_streams_vs_oracle, repeated; reports where HIP and the oracle differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import async_ev_cnn_amd as A
from async_ev_cnn_amd import _native as N
from oracle import oracle as O
from tests.util import build_chain, uniform_steps
from tests.test_gpu_parity import random_tile_architecture

seed, reps = int(sys.argv[1]), int(sys.argv[2])
for rep in range(reps):
    rs, spec, h, w = random_tile_architecture(seed)
    S = int(rs.choice([3, 17, 70, 130])); T = 4
    os.environ["AEV_TUNING"] = str(int(rs.choice([0, 0, 1, 2])))
    if rs.rand() < 0.3:
        os.environ["AEV_FORCE_QUEUE"] = "1"
    names, nc = [], 0
    for sp in spec:
        if sp[0] == "conv":
            nc += 1; names.append("conv%d=%d,%d,%d,%d" % (nc, sp[1], sp[2], sp[3], sp[4]))
        else:
            names.append("pool%d=%d,%d" % (nc, sp[1], sp[1]))
    layers = A.layers_dict(" ".join(names))
    wts = A.net.he_init_weights(layers, 50 + seed)
    wl = [(wts['w_' + n], wts['b_' + n]) for n in layers if 'conv' in n]
    leak = 10.0 ** -rs.randint(2, 5)
    per = [uniform_steps(np.random.RandomState(9000 + 131 * seed + s), h, w, T, 1 + int(rs.randint(0, 14)), dt_hi=int(rs.randint(5, 300))) for s in range(S)]
    for _ in range(max(1, S // 8)):
        per[int(rs.randint(0, S))][int(rs.randint(0, T))] = np.zeros((0, 3), np.int32)
    steps = [[per[s][t] for s in range(S)] for t in range(T)]
    net = A.EventNet(h, w, layers, wts, 'SAME', 0.1, leak, S)
    orcs = [build_chain(O, spec, wl, h, w, leak, 0.1, "SAME") for s in range(S)]
    bad0 = [(s, li) for s in range(S) for li in range(len(orcs[s])) if not np.array_equal(net.read(li, N.FEATUREMAP, s), orcs[s][li].featuremap())]
    net.queue_upload(steps)
    graph = bool(rs.rand() < 0.7)
    net.queue_run(0, 1, use_graph=graph); net.queue_run(1, T - 1, use_graph=graph)
    for s in range(S):
        for t in range(T):
            if len(steps[t][s]):
                O.net_step(orcs[s][-1], steps[t][s])
    nbad = 0
    for s in range(S):
        for li in range(len(orcs[s])):
            g, r = net.read(li, N.FEATUREMAP, s), orcs[s][li].featuremap()
            if not np.array_equal(g, r):
                d = np.argwhere(g != r)
                ys, xs = np.unique(d[:, 1]), np.unique(d[:, 2])
                print("rep %d (graph %s tuning %s): stream %d layer %d: %d values differ, channels %d..%d, rows %s, cols %s; initial-state mismatches %s" % (
                    rep, graph, os.environ["AEV_TUNING"], s, li, len(d), d[:, 0].min(), d[:, 0].max(), ys.tolist()[:12], xs.tolist()[:12], bad0[:4]))
                nbad += 1
                break
    if rep == 0:
        print("seed", seed, spec, h, w, "S", S, "tuning", os.environ["AEV_TUNING"], "force_queue", os.environ.get("AEV_FORCE_QUEUE"), "graph", graph, flush=True)
    del net
print("done")
