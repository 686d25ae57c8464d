"""Feasibility numbers for a BLOCKED exact decay: in a queued run the next K steps' delta_leak are known, and a site whose
|surface| exceeds |decay| x their sum cannot change sign in any of them -- it could skip K - 1 sweeps and have the K subtractions replayed, in
order, by its next reader.  How many of the sites the sweep visits today (decay != 0) are that safe?  e-fcn, N-Caltech101-like streams, after
300 steps, per conv layer of the look-ahead (conv1 ... conv4) and K = 2 ... 16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import async_ev_cnn_amd as A
from async_ev_cnn_amd import _native as N
import bench

S, T0, KMAX, LEAK = 4, 300, 16, 5e-5
layers = A.layers_dict(A.net.EFCN_LAYERS)
net = A.EventNet(160, 224, layers, A.net.he_init_weights(layers, 1234), 'SAME', 0.1, LEAK, S)
per = [bench.make_stream("ncaltech", 1234 + s, T0 + KMAX, 200) for s in range(S)]
net.queue_upload([[per[s][t] for s in range(S)] for t in range(T0 + KMAX)])
net.queue_run(0, T0, use_graph=True)
last = lambda s, t: float(per[s][t][-1, 2])
convs = [li for li, k in enumerate(net.kinds) if k == "conv"][:4]
tot_bytes, saved = 0.0, {K: 0.0 for K in (2, 4, 8, 16)}
for li in convs:
    row = {K: [] for K in saved}
    nz_share = []
    for s in range(S):
        f = np.abs(net.read(li, N.SURFACE, s).astype(np.float64))
        a = np.abs(net.read(li, N.RAW_ACT, s).astype(np.float64))
        site_nz = (a != 0).any(axis=0)                     # the sweep skips a SITE whose decay is zero in every channel
        nz_share.append(site_nz.mean())
        for K in saved:
            lam = (last(s, T0 + K - 1) - last(s, T0 - 1)) * LEAK
            safe = ((f > 1.001 * a * lam) | (a == 0)).all(axis=0)   # every channel of the site
            row[K].append((safe & site_nz).sum() / max(site_nz.sum(), 1))
    nbytes = 12.0 * net.read(li, N.SURFACE, 0).size * np.mean(nz_share)   # what the sweep moves for this layer per stream-step
    tot_bytes += nbytes
    print("layer %d: sites with decay %.2f; of those, safe for K steps: %s" % (li, np.mean(nz_share), "  ".join("K=%d %.3f" % (K, np.mean(row[K])) for K in saved)))
    for K in saved:
        saved[K] += nbytes * np.mean(row[K]) * (1 - 1.0 / K)
print("sweep bytes of conv1..conv4 per stream-step today %.2f MB; a blocked decay would not move: %s" %
      (tot_bytes / 1e6, "  ".join("K=%d %.0f %%" % (K, 100 * saved[K] / tot_bytes) for K in saved)))
