"""GPU fuzz: seeded random architectures / stream counts / tunings, whole-chain stepping (direct, queue, graph) against the
oracle, bit for bit.  Exercises the kernel-configuration logic (tile sizes that do not divide the map, channel counts that are
not multiples of anything, 1x1 and 3x3 mixes, pools on odd positions in the chain, VALID and SAME)."""
import numpy as np
import pytest

import async_ev_cnn_amd as A
from async_ev_cnn_amd import _native as N
from oracle import oracle as O
from tests.util import build_chain, describe_mismatch, he_weights, int_weights, uniform_steps

pytestmark = pytest.mark.gpu

CHANNELS = [1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 33, 40, 48, 64, 70, 96, 128, 130, 200, 256]


def random_net(rs):
    for _ in range(100):
        h, w = int(rs.choice([8, 10, 12, 14, 16, 20, 24, 28, 32, 36, 40])), int(rs.choice([8, 10, 12, 14, 16, 20, 24, 28, 32, 44]))
        padding = "SAME" if rs.rand() < 0.8 else "VALID"
        spec, cin, ch, cw, prev_pool = [], 1, h, w, True
        nl = rs.randint(1, 6)
        for li in range(nl):
            if not prev_pool and rs.rand() < 0.35 and ch % 2 == 0 and cw % 2 == 0 and ch >= 4 and cw >= 4:
                spec.append(("pool", 2)); ch //= 2; cw //= 2; prev_pool = True
                continue
            k = 3 if (rs.rand() < 0.7 or li == 0) else 1
            if padding == "VALID" and k == 3:
                if ch < 5 or cw < 5:
                    break
                ch -= 2; cw -= 2
            cout = int(rs.choice(CHANNELS[1:] if li else CHANNELS[1:8]))
            spec.append(("conv", k, k, cin, cout)); cin = cout; prev_pool = False
        if spec and any(s[0] == "conv" for s in spec):
            return h, w, padding, spec
    raise RuntimeError("no net")


@pytest.mark.parametrize("seed", range(96))
def test_random_architecture_vs_oracle(seed):
    rs = np.random.RandomState(1000 + seed)
    h, w, padding, spec = random_net(rs)
    S = int(rs.choice([1, 1, 2, 3, 5]))
    tuning = ["auto", "throughput", "latency"][seed % 3]
    exact = seed % 4 == 0
    wl = int_weights(rs, spec, -1, 2) if exact else he_weights(rs, spec)
    names, wd, ci, pi = [], {}, 0, 0
    for sp in spec:
        if sp[0] == "conv":
            ci += 1
            names.append("conv%d=%d,%d,%d,%d" % (ci, sp[1], sp[2], sp[3], sp[4]))
            wd["w_conv%d" % ci], wd["b_conv%d" % ci] = wl[ci - 1]
        else:
            pi += 1
            names.append("pool%d=2,2" % pi)
    leak, alpha = (2.0 ** -5, 0.125) if exact else (1e-3, 0.1)
    net = A.EventNet(h, w, A.layers_dict(" ".join(names)), wd, padding, alpha, leak, S, tuning=tuning)
    orcs = [build_chain(O, spec, wl, h, w, leak, alpha, padding) for _ in range(S)]
    T = 6
    per = [uniform_steps(np.random.RandomState(seed * 10 + s), h, w, T, int(rs.randint(1, 25)), dt_hi=40) for s in range(S)]
    if S > 1:
        per[1][2] = np.zeros((0, 3), np.int32)           # one idle stream-step
    steps = [[per[s][t] for s in range(S)] for t in range(T)]
    if seed % 2:
        net.queue_upload(steps)
        net.queue_run(0, 3, use_graph=True)
        net.queue_run(3, T - 3, use_graph=bool(seed % 3))
    else:
        for t in range(T):
            net.step(steps[t])
    for s in range(S):
        for t in range(T):
            if len(per[s][t]):
                O.net_step(orcs[s][-1], per[s][t])
        for li, lo in enumerate(orcs[s]):
            tag = (seed, spec, S, tuning, s, li)
            got = net.read(li, N.FEATUREMAP, s)
            if not np.array_equal(got, lo.featuremap()):
                fresh = build_chain(O, spec, wl, h, w, leak, alpha, padding)
                raise AssertionError("%s h %d w %d %s: %s" % (tag, h, w, padding, describe_mismatch(got, lo.featuremap(), fresh[li].featuremap())))
            assert np.array_equal(net.read(li, N.CONV_ACTFN, s), lo.conv_actfn()), tag
            if hasattr(lo, "idx_max"):
                assert np.array_equal(net.read(li, N.IDX_MAX, s), lo.idx_max()), tag
                assert np.array_equal(net.read(li, N.RECOMPUTE, s).astype(bool), lo.recompute_coords()), tag
            if len(per[s][T - 1]):
                y, x = net.layer_events(li, s)
                oy, ox = lo._events()
                assert np.array_equal(y, oy) and np.array_equal(x, ox), tag
