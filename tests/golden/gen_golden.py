#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and oracle/_ref built by
oracle/build_ref.sh).  It imports the reference's own, unmodified layer classes
(src/layers/{integration,conv2d,maxpool}.py) and its Cython ops (src/libs/cutils.pyx),
drives them with seeded inputs, and stores inputs + expected outputs as small .npz
files.  Only data is written: no reference source text ends up in the repository.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [--timing]

Fixture families:
  exact_*  dyadic leak / alpha, small-integer weights and timestamps: every value is
           exactly representable, so float32 (the contract) and the float64 that the
           reference slips into under NumPy >= 2 agree bit for bit -> bit-exact checks of
           values AND of all event / index bookkeeping.
  float_*  He-init float weights or leak 0.1: compared at atol 1e-5, event sets compared
           only where |value| > 1e-4.
Each file records the NumPy version that produced it.
"""
import argparse
import json
import os
import sys
import time

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AEV_REFERENCE_DIR", "/root/reference")


def import_reference():
    sys.path.insert(0, REF)
    import src.libs  # noqa

    src.libs.__path__.append(os.path.join(REPO, "oracle", "_ref", "src", "libs"))
    from src.libs.cutils import im2col_event, min_argmax
    from src.layers.integration import IntegrationLayer
    from src.layers.conv2d import Conv2DLayer
    from src.layers.maxpool import MaxPoolLayer

    return im2col_event, min_argmax, IntegrationLayer, Conv2DLayer, MaxPoolLayer


# ----------------------------------------------------------------------------- op KATs
def gen_ops(im2col_event, min_argmax):
    rs = np.random.RandomState(7)
    out = {}
    cases = []

    def add_im2col(name, img, ey, ex, kh, kw, stride, cac):
        cols, (oy, ox) = im2col_event(img, ey.astype(np.int32), ex.astype(np.int32), kh, kw, stride, cac)
        out[name + "_img"] = img
        out[name + "_ey"] = ey.astype(np.int32)
        out[name + "_ex"] = ex.astype(np.int32)
        out[name + "_cols"] = np.asfortranarray(cols)
        out[name + "_oy"] = oy
        out[name + "_ox"] = ox
        cases.append({"name": name, "kh": kh, "kw": kw, "stride": stride, "chan_as_cols": cac})

    img6 = np.arange(36, dtype=np.float32).reshape(1, 6, 6)
    add_im2col("survey_kat", img6, np.array([3, 3, 0]), np.array([3, 4, 0]), 3, 3, 1, 0)
    add_im2col("corners", img6, np.array([0, 0, 5, 5, 2]), np.array([0, 5, 0, 5, 2]), 3, 3, 1, 0)
    add_im2col("dups", img6, np.array([2, 2, 2, 3, 2]), np.array([2, 2, 3, 3, 2]), 3, 3, 1, 0)
    add_im2col("k1", rs.randn(3, 5, 7).astype(np.float32), rs.randint(0, 5, 9), rs.randint(0, 7, 9), 1, 1, 1, 0)
    for c in (1, 2, 16):
        img = rs.randn(c, 10, 12).astype(np.float32)
        ey, ex = rs.randint(0, 10, 25), rs.randint(0, 12, 25)
        add_im2col("c%d_k3" % c, img, ey, ex, 3, 3, 1, 0)
        add_im2col("c%d_k3_cac" % c, img, ey, ex, 3, 3, 1, 1)
        add_im2col("c%d_pool" % c, img, ey, ex, 2, 2, 2, 0)
        add_im2col("c%d_pool_cac" % c, img, ey, ex, 2, 2, 2, 1)
    add_im2col("empty", img6, np.zeros(0, np.int64), np.zeros(0, np.int64), 3, 3, 1, 0)
    add_im2col("rect_k", rs.randn(2, 9, 8).astype(np.float32), rs.randint(0, 9, 12), rs.randint(0, 8, 12), 3, 2, 1, 0)
    add_im2col("full", rs.randn(2, 8, 8).astype(np.float32), np.repeat(np.arange(8), 8), np.tile(np.arange(8), 8), 3, 3, 1, 0)

    mcases = []

    def add_minarg(name, mx, mn):
        mx, mn = np.asfortranarray(mx, dtype=np.float32), np.asfortranarray(mn, dtype=np.float32)
        am, na = min_argmax(mx, mn)
        out[name + "_mx"], out[name + "_mn"], out[name + "_am"], out[name + "_na"] = mx, mn, am, na
        mcases.append(name)

    add_minarg("ma_survey", np.array([[1, 5, 5, 2]]).T, np.array([[0, 3, 1, 0]]).T)
    add_minarg("ma_ties", np.array([[1, 1, 1, 1], [2, 2, 0, 2], [0, 3, 3, 3], [-1, -1, -1, -1]]).T,
               np.array([[4, 3, 2, 1], [1, 1, 0, 0], [0, 2, 2, 1], [0, 0, 0, 0]]).T)
    mx = rs.randint(-2, 3, size=(4, 400)).astype(np.float32)
    mn = rs.randint(-2, 3, size=(4, 400)).astype(np.float32)
    add_minarg("ma_rand_int", mx, mn)
    add_minarg("ma_rand_f", rs.randn(4, 300), rs.randn(4, 300))
    add_minarg("ma_r9", rs.randint(-1, 2, size=(9, 100)), rs.randint(-1, 2, size=(9, 100)))
    out["meta"] = np.array(json.dumps({"im2col": cases, "min_argmax": mcases, "numpy": np.__version__}))
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **out)
    print("ops.npz: %d im2col cases, %d min_argmax cases" % (len(cases), len(mcases)))


# ----------------------------------------------------------------------------- layer traces
def build_chain(classes, spec, weights, h, w, leak, alpha, padding):
    """spec: list of ('conv', kh, kw, cin, cout) / ('pool', k).  Mirrors event_numpy.py:53-73."""
    _, _, IntegrationLayer, Conv2DLayer, MaxPoolLayer = classes
    layers = [IntegrationLayer(leak, h, w)]
    wi = 0
    for s in spec:
        if s[0] == "conv":
            k, b = weights[wi]
            wi += 1
            layers.append(Conv2DLayer(layers[-1], k, b, 1, alpha, padding))
        else:
            layers.append(MaxPoolLayer(layers[-1], [s[1], s[1]], s[1]))
    return layers


def run_trace(name, classes, spec, weights, h, w, leak, alpha, padding, steps, snap_steps, snap_layers=None,
              family="exact", reset_at=None, weights_rule=None):
    layers = build_chain(classes, spec, weights, h, w, leak, alpha, padding)
    nl = len(layers)
    snap_layers = list(range(nl)) if snap_layers is None else snap_layers
    ev_lists = [[] for _ in range(nl)]
    sums = [[] for _ in range(nl)]
    deltas = []
    snaps = {l: {"fm": [], "act": [], "surf": [], "idx": [], "pend": []} for l in snap_layers}
    for t, ev in enumerate(steps):
        if reset_at is not None and t == reset_at:
            for l in layers:
                l.reset()
        evs, dl = layers[0].compute(ev, None)
        ev_lists[0].append(np.stack(evs, -1).astype(np.int32))
        for i in range(1, nl):
            evs, dl = layers[i].compute(evs, dl)
            ev_lists[i].append(np.stack(evs, -1).astype(np.int32).reshape(-1, 2))
        deltas.append(float(dl))
        for i in range(nl):
            sums[i].append(float(np.sum(layers[i].featuremap(), dtype=np.float64)))
        if t in snap_steps:
            for l in snap_layers:
                L = layers[l]
                snaps[l]["fm"].append(np.array(L.featuremap(), dtype=np.float32, copy=True))
                snaps[l]["act"].append(np.array(L.conv_actfn(), dtype=np.float32, copy=True))
                snaps[l]["surf"].append(np.array(L.surface(), dtype=np.float32, copy=True))  # live array: copy
                if hasattr(L, "_idx_max"):
                    snaps[l]["idx"].append(L._idx_max[0].astype(np.int32).copy())
                    snaps[l]["pend"].append(L._recompute_coords.copy())
    out = {}
    all_ev = np.concatenate(steps, 0).astype(np.int32)
    out["events"] = all_ev
    out["events_off"] = np.cumsum([0] + [len(s) for s in steps]).astype(np.int64)
    for i in range(nl):
        cat = np.concatenate(ev_lists[i], 0) if ev_lists[i] else np.zeros((0, 2), np.int32)
        out["ev%d" % i] = cat.astype(np.int16)
        out["ev%d_off" % i] = np.cumsum([0] + [len(e) for e in ev_lists[i]]).astype(np.int64)
        out["sum%d" % i] = np.array(sums[i], np.float64)
    out["delta"] = np.array(deltas, np.float64)
    out["snap_steps"] = np.array(sorted(snap_steps), np.int64)
    for l in snap_layers:
        for k, v in snaps[l].items():
            if v:
                out["%s%d" % (k, l)] = np.stack(v, 0)
    if weights_rule is None:  # big nets: weights are re-drawn from the seeded rule instead of being stored
        for wi, (k, b) in enumerate(weights):
            out["w%d" % wi] = np.asarray(k, np.float32)
            out["b%d" % wi] = np.asarray(b, np.float32).reshape(-1)
    out["meta"] = np.array(json.dumps({
        "name": name, "family": family, "spec": spec, "h": h, "w": w, "leak": leak, "alpha": alpha,
        "padding": padding, "n_steps": len(steps), "snap_layers": snap_layers, "reset_at": reset_at,
        "weights_rule": weights_rule,
        "numpy": np.__version__, "shapes": [list(map(int, l.out_shape())) for l in layers]}))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: %d steps, %.1f KB" % (name, len(steps), os.path.getsize(path) / 1024.0))


def event_schedule(rs, h, w, n_steps, per_step, first_single=True, dt_lo=1, dt_hi=10):
    """Stimulus of test_correctness.py:121-126,165-168 with a fixed RandomState."""
    steps, prev = [], 0
    for t in range(n_steps):
        n = 1 if (t == 0 and first_single) else per_step
        ts = (np.zeros(1, np.int64) if (t == 0 and first_single) else np.sort(rs.randint(dt_lo, dt_hi, size=n)) + prev)
        y, x = rs.randint(0, h, size=n), rs.randint(0, w, size=n)
        steps.append(np.stack([y, x, ts], -1).astype(np.int32))
        prev = int(ts.max())
    return steps


def he_weights(rs, spec):
    """Weights rule W of SURVEY.md 8d."""
    ws = []
    for s in spec:
        if s[0] == "conv":
            _, kh, kw, ci, co = s
            k = (rs.randn(kh, kw, ci, co) * np.sqrt(2.0 / (kh * kw * ci))).astype(np.float32)
            b = (rs.randn(co) * 0.1).astype(np.float32)
            ws.append((k, b))
    return ws


def int_weights(rs, spec, lo=-2, hi=3):
    ws = []
    for s in spec:
        if s[0] == "conv":
            _, kh, kw, ci, co = s
            ws.append((rs.randint(lo, hi, size=(kh, kw, ci, co)).astype(np.float32),
                       rs.randint(-3, 4, size=co).astype(np.float32)))
    return ws


EFCN = [("conv", 3, 3, 1, 16), ("pool", 2), ("conv", 3, 3, 16, 32), ("pool", 2), ("conv", 3, 3, 32, 64), ("pool", 2),
        ("conv", 3, 3, 64, 128), ("pool", 2), ("conv", 3, 3, 128, 256), ("pool", 2), ("conv", 1, 1, 256, 512),
        ("conv", 1, 1, 512, 110)]


def gen_layers(classes):
    # (1) the literal net of test_correctness.py:96-105: 8x8, leak .1, alpha .1, kernel rows [-2,-1,1], bias 10
    spec = [("conv", 3, 3, 1, 1), ("pool", 2), ("conv", 3, 3, 1, 1), ("pool", 2)]
    k = np.array([[-2, -1, 1]] * 3).reshape(3, 3, 1, 1)
    lit_w = [(k, np.array([10])), (k, np.array([10]))]
    run_trace("float_testnet8", classes, spec, lit_w, 8, 8, 0.1, 0.1, "SAME",
              event_schedule(np.random.RandomState(0), 8, 8, 200, 5), {0, 1, 2, 10, 50, 199}, family="float")
    # same net, dyadic leak/alpha -> exact
    run_trace("exact_testnet8", classes, spec, lit_w, 8, 8, 0.0625, 0.125, "SAME",
              event_schedule(np.random.RandomState(1), 8, 8, 300, 5), {0, 1, 2, 3, 10, 50, 150, 299}, family="exact")
    # (2) config-1 nets: 64x64, one 3x3 conv 1->1 and 1->16, 1000 events, 1 and 5 events per step
    for co in (1, 16):
        sp = [("conv", 3, 3, 1, co)]
        rs = np.random.RandomState(0)
        wi = int_weights(rs, sp)
        run_trace("exact_cfg1_c%d_e1" % co, classes, sp, wi, 64, 64, 2.0 ** -6, 0.125, "SAME",
                  event_schedule(rs, 64, 64, 1000, 1), {0, 99, 999}, family="exact")
        run_trace("exact_cfg1_c%d_e5" % co, classes, sp, wi, 64, 64, 2.0 ** -6, 0.125, "SAME",
                  event_schedule(rs, 64, 64, 200, 5), {0, 99, 199}, family="exact")
    sp = [("conv", 3, 3, 1, 16)]
    rs = np.random.RandomState(1234)
    run_trace("float_cfg1_c16_e5", classes, sp, he_weights(rs, sp), 64, 64, 5e-5, 0.1, "SAME",
              event_schedule(rs, 64, 64, 200, 5, dt_lo=1, dt_hi=2000), {0, 99, 199}, family="float")
    # (3) 32x32 multi-channel net with a 1x1 conv, exact and float; VALID variant; reset in the middle
    sp = [("conv", 3, 3, 1, 4), ("pool", 2), ("conv", 3, 3, 4, 8), ("pool", 2), ("conv", 1, 1, 8, 6)]
    rs = np.random.RandomState(3)
    run_trace("exact_net32", classes, sp, int_weights(rs, sp, -1, 2), 32, 32, 2.0 ** -5, 0.125, "SAME",
              event_schedule(rs, 32, 32, 120, 8), {0, 1, 5, 30, 119}, family="exact")
    rs = np.random.RandomState(4)
    run_trace("exact_net32_reset", classes, sp, int_weights(rs, sp, -1, 2), 32, 32, 2.0 ** -5, 0.125, "SAME",
              event_schedule(rs, 32, 32, 40, 8, first_single=False), {0, 19, 20, 39}, family="exact", reset_at=20)
    rs = np.random.RandomState(5)
    run_trace("float_net32", classes, sp, he_weights(rs, sp), 32, 32, 1e-3, 0.1, "SAME",
              event_schedule(rs, 32, 32, 150, 10, dt_lo=1, dt_hi=40), {0, 1, 5, 50, 149}, family="float")
    spv = [("conv", 3, 3, 1, 3), ("pool", 2), ("conv", 3, 3, 3, 5)]
    rs = np.random.RandomState(6)
    run_trace("exact_net34_valid", classes, spv, int_weights(rs, spv, -1, 2), 34, 34, 2.0 ** -5, 0.25, "VALID",
              event_schedule(rs, 34, 34, 80, 6), {0, 1, 40, 79}, family="exact")
    # (4) reduced-step e-fcn (configs/efcn_event.yml:6-18): 160x224, B=200, weights rule W, seed 1234
    rs = np.random.RandomState(1234)
    w_efcn = he_weights(rs, EFCN)
    steps, prev = [], 0
    for t in range(3):
        ts = np.sort(rs.randint(0, 1000, 200)) + prev
        steps.append(np.stack([rs.randint(0, 160, 200), rs.randint(0, 224, 200), ts], -1).astype(np.int32))
        prev = int(ts.max())
    run_trace("float_efcn3", classes, EFCN, w_efcn, 160, 224, 5e-5, 0.1, "SAME", steps, {2},
              snap_layers=[9, 11, 12], family="float", weights_rule={"rule": "he_weights", "seed": 1234})


def gen_pool_stacks(classes):
    """Round 4: chains with a pool layer directly behind a pool layer (MaxPoolLayer takes any prev_layer, maxpool.py:14-40): the
    reference's own layer classes, exact family (small-integer weights, dyadic leak / alpha) and one float trace."""
    sp = [("conv", 3, 3, 1, 4), ("pool", 2), ("pool", 2), ("conv", 3, 3, 4, 8), ("pool", 2), ("pool", 2), ("conv", 1, 1, 8, 6)]
    rs = np.random.RandomState(41)
    run_trace("exact_poolstack32", classes, sp, int_weights(rs, sp, -1, 2), 32, 32, 2.0 ** -5, 0.125, "SAME",
              event_schedule(rs, 32, 32, 100, 8), {0, 1, 5, 30, 99}, family="exact")
    rs = np.random.RandomState(42)
    run_trace("exact_poolstack32_reset", classes, sp, int_weights(rs, sp, -1, 2), 32, 32, 2.0 ** -5, 0.125, "SAME",
              event_schedule(rs, 32, 32, 40, 8, first_single=False), {0, 19, 20, 39}, family="exact", reset_at=20)
    sp3 = [("conv", 3, 3, 1, 8), ("pool", 2), ("pool", 2), ("pool", 2)]          # three pool layers in a row
    rs = np.random.RandomState(43)
    run_trace("exact_poolstack_triple", classes, sp3, int_weights(rs, sp3, -1, 2), 32, 32, 2.0 ** -5, 0.25, "SAME",
              event_schedule(rs, 32, 32, 60, 6), {0, 1, 30, 59}, family="exact")
    rs = np.random.RandomState(44)
    run_trace("float_poolstack32", classes, sp, he_weights(rs, sp), 32, 32, 1e-3, 0.1, "SAME",
              event_schedule(rs, 32, 32, 100, 10, dt_lo=1, dt_hi=40), {0, 1, 5, 50, 99}, family="float")


def gen_nms():
    """apply_nms of the reference (src/libs/utils.py:38-118) on seeded boxes: inputs + picked indices."""
    sys.path.insert(0, REF)
    from src.libs.utils import apply_nms
    rs = np.random.RandomState(11)
    out, cases = {}, []
    for ci, (nbx, thr, pvalid) in enumerate([(70, 0.0, 0.5), (70, 0.2, 0.7), (70, 0.5, 1.0), (12, 0.3, 0.6), (200, 0.4, 0.4),
                                             (70, 0.2, 0.0)]):
        b = np.stack([rs.uniform(0, 224, (3, nbx)), rs.uniform(0, 160, (3, nbx)), rs.uniform(2, 90, (3, nbx)),
                      rs.uniform(2, 90, (3, nbx))], -1).astype(np.float32)
        sc = rs.uniform(0, 1, (3, nbx)).astype(np.float32)
        valid = rs.uniform(0, 1, (3, nbx)) < pvalid
        if valid.any():
            i0, i1 = apply_nms(b, sc, valid, iou_threshold=thr)
        else:
            i0, i1 = np.zeros(0, np.int64), np.zeros(0, np.int64)
        n = "nms%d" % ci
        out[n + "_b"], out[n + "_s"], out[n + "_v"] = b, sc, valid
        out[n + "_i0"], out[n + "_i1"] = np.asarray(i0, np.int64), np.asarray(i1, np.int64)
        cases.append({"name": n, "thr": thr})
    out["meta"] = np.array(json.dumps({"cases": cases, "numpy": np.__version__}))
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **out)
    print("nms.npz: %d cases" % len(cases))


def _absent_module(name, **attrs):
    """An EMPTY placeholder for a third-party module that is not installed here and that the reference function under
    test never touches (it only sits in the import list of the file that holds the function).  No functionality is
    provided: any use of it raises AttributeError."""
    import types
    if name in sys.modules:
        return
    try:
        __import__(name)
    except ImportError:
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m


def gen_nformat():
    """NReader.read_example of the reference (src/readers/file_reader.py:30-58, pure NumPy) on seeded byte streams written
    to temporary files: raw bytes + the (length, x, y, ts, p) it returns.  The file's unused imports of `bitstring` and
    `numba` (neither installed, neither touched by NReader) are satisfied by empty placeholder modules."""
    import tempfile
    sys.path.insert(0, REF)
    _absent_module("bitstring")
    _absent_module("numba", njit=None)
    from src.readers.file_reader import NReader
    rd = NReader()
    rs = np.random.RandomState(5)
    out, cases = {}, []

    def encode(x, y, ts, p, ovf_at):
        rec = []
        carry = 0
        for i in range(len(x)):
            while ovf_at and i in ovf_at[carry:carry + 1]:
                rec.append([rs.randint(0, 240), 240, rs.randint(0, 256), rs.randint(0, 256), rs.randint(0, 256)])  # marker: y == 240
                carry += 1
            t = int(ts[i]) - (carry << 13)
            rec.append([x[i], y[i], (p[i] << 7) | ((t >> 16) & 127), (t >> 8) & 255, t & 255])
        return np.array(rec, np.uint8).reshape(-1)

    specs = [("plain", 500, []), ("markers", 3000, sorted(rs.choice(np.arange(1, 3000), 40, replace=False).tolist())),
             ("marker_first_last", 64, [0, 63]), ("one_event", 1, []), ("markers_only", 0, [])]
    for name, nev, ovf in specs:
        x, y = rs.randint(0, 240, nev), rs.randint(0, 180, nev)
        p = rs.randint(0, 2, nev)
        ts = np.sort(rs.randint(0, 1 << 20, nev)) + (len(ovf) << 13)
        raw = encode(x, y, ts, p, ovf) if nev else np.array([[7, 240, 1, 2, 3], [9, 240, 0, 0, 1]], np.uint8).reshape(-1)
        with tempfile.NamedTemporaryFile(suffix=".bin") as f:
            f.write(raw.tobytes()); f.flush()
            length, rx, ry, rts, rp = rd.read_example(f.name)
        out[name + "_raw"] = raw
        out[name + "_x"], out[name + "_y"], out[name + "_ts"], out[name + "_p"] = rx, ry, rts, rp
        cases.append({"name": name, "length": int(length)})
    out["meta"] = np.array(json.dumps({"cases": cases, "numpy": np.__version__}))
    np.savez_compressed(os.path.join(HERE, "nformat.npz"), **out)
    print("nformat.npz: %d cases" % len(cases))


def gen_bbox_decode():
    """convert_bboxes of the reference (src/libs/viz.py:27-46, pure NumPy) on seeded prediction tensors, with the arguments
    draw_bboxes passes (viz.py:136-140: sqrt=True).  viz.py's top-level `import cv2` (not installed; used by the drawing
    functions only) is satisfied by an empty placeholder module."""
    sys.path.insert(0, REF)
    _absent_module("cv2")
    from src.libs.viz import convert_bboxes
    rs = np.random.RandomState(6)
    out, cases = {}, []
    for ci, (gh, gw, C_, B, hi, wi) in enumerate([(5, 7, 100, 2, 160, 224), (3, 4, 5, 3, 96, 128), (1, 1, 2, 1, 32, 32)]):
        pred = rs.uniform(-0.5, 1.5, (gh, gw, C_ + 5 * B)).astype(np.float32)
        bb = np.reshape(pred[None, ..., C_:], [1, gh, gw, -1, 5])[..., 0:4]                    # viz.py:133-136
        boxes = convert_bboxes(bb, gh, gw, hi, wi, sqrt=True)                                   # viz.py:138
        n = "dec%d" % ci
        out[n + "_pred"], out[n + "_boxes"] = pred, np.reshape(boxes, [-1, 4])                  # viz.py:144 flattening
        cases.append({"name": n, "gh": gh, "gw": gw, "C": C_, "B": B, "h_image": hi, "w_image": wi, "dtype": str(boxes.dtype)})
    out["meta"] = np.array(json.dumps({"cases": cases, "numpy": np.__version__}))
    np.savez_compressed(os.path.join(HERE, "bbox_decode.npz"), **out)
    print("bbox_decode.npz: %d cases" % len(cases))


# ----------------------------------------------------------------------------- round 3: the free functions, the feeder, the dense model
def gen_dense_ops():
    """conv2d_event / conv2d (src/layers/conv2d.py:144-230) and the dense im2col (src/layers/functional.py:4-34) of the
    reference on seeded inputs.  `exact` cases hold small integers (any summation order gives the same bits), `float` cases
    are compared at 1e-5."""
    import_reference()
    from src.layers.conv2d import conv2d, conv2d_event
    from src.layers.functional import im2col
    rs = np.random.RandomState(31)
    out, ce, cd, ic = {}, [], [], []

    def add_ce(name, img, ey, ex, kernel, bias, stride=1, exact=True):
        conv, (oy, ox) = conv2d_event(img, (ey, ex), kernel, bias, stride=stride)
        out[name + "_img"], out[name + "_ey"], out[name + "_ex"], out[name + "_k"] = img, ey, ex, kernel
        if bias is not None:
            out[name + "_b"] = bias
        out[name + "_conv"], out[name + "_oy"], out[name + "_ox"] = conv, oy, ox
        ce.append({"name": name, "stride": stride, "bias": bias is not None, "exact": exact, "dtype": str(conv.dtype)})

    ints = lambda *shape: rs.randint(-3, 4, size=shape).astype(np.float32)
    add_ce("ce_c1", ints(1, 10, 12), rs.randint(0, 10, 15), rs.randint(0, 12, 15), ints(4, 1, 3, 3), None)
    add_ce("ce_c8_bias32", ints(8, 12, 14), rs.randint(0, 12, 30), rs.randint(0, 14, 30), ints(16, 8, 3, 3), ints(16))
    add_ce("ce_bias64", ints(3, 9, 9), rs.randint(0, 9, 12), rs.randint(0, 9, 12), ints(5, 3, 3, 3), rs.randint(-3, 4, 5).astype(np.float64))
    add_ce("ce_bias_int", ints(1, 8, 8), rs.randint(0, 8, 5), rs.randint(0, 8, 5), ints(1, 1, 3, 3), np.array([10]))   # test_correctness.py:104
    add_ce("ce_k1", ints(5, 7, 9), rs.randint(0, 7, 20), rs.randint(0, 9, 20), ints(6, 5, 1, 1), ints(6))
    add_ce("ce_pool_stride", ints(4, 8, 12), rs.randint(0, 8, 10), rs.randint(0, 12, 10), ints(3, 4, 2, 2), None, stride=2)
    add_ce("ce_int64_events_f64_image", ints(2, 9, 9).astype(np.float64), rs.randint(0, 9, 11).astype(np.int64), rs.randint(0, 9, 11).astype(np.int64),
           ints(3, 2, 3, 3), None)
    add_ce("ce_empty", ints(2, 8, 8), np.zeros(0, np.int64), np.zeros(0, np.int64), ints(3, 2, 3, 3), ints(3))
    add_ce("ce_float_c16", rs.randn(16, 22, 30).astype(np.float32), rs.randint(0, 22, 120), rs.randint(0, 30, 120),
           (rs.randn(32, 16, 3, 3) / 12).astype(np.float32), (rs.randn(32) * 0.1).astype(np.float32), exact=False)

    def add_cd(name, img, kernel, bias, padding, stride, exact=True):
        conv = conv2d(img, kernel, bias, padding, stride)
        out[name + "_img"], out[name + "_k"], out[name + "_out"] = img, kernel, conv
        if bias is not None:
            out[name + "_b"] = bias
        cd.append({"name": name, "padding": padding, "stride": stride, "bias": bias is not None, "exact": exact, "dtype": str(conv.dtype)})

    add_cd("cd_valid", ints(3, 9, 11), ints(5, 3, 3, 3), ints(5), 'VALID', 1)
    add_cd("cd_same", ints(3, 9, 11), ints(5, 3, 3, 3), None, 'SAME', 1)
    add_cd("cd_same_s2", ints(2, 9, 11), ints(4, 2, 3, 3), ints(4), 'SAME', 2)
    add_cd("cd_same_s2_even", ints(2, 10, 12), ints(4, 2, 3, 3), None, 'SAME', 2)
    add_cd("cd_same_k5", ints(2, 8, 8), ints(3, 2, 5, 5), None, 'SAME', 1)
    add_cd("cd_valid_s3", ints(2, 11, 13), ints(3, 2, 2, 2), None, 'VALID', 3)
    add_cd("cd_batch_valid", ints(3, 2, 8, 8), ints(4, 2, 3, 3), ints(4), 'VALID', 1)
    add_cd("cd_q3_padding_slot_holds_the_stride", ints(1, 8, 8), ints(2, 1, 3, 3), ints(2), 1, 1)   # conv2d.py:59-60
    add_cd("cd_bias64", ints(2, 7, 7), ints(3, 2, 3, 3), rs.randint(-3, 4, 3).astype(np.float64), 'SAME', 1)
    add_cd("cd_k1", ints(6, 5, 7), ints(9, 6, 1, 1), ints(9), 'SAME', 1)
    add_cd("cd_float", rs.randn(16, 20, 28).astype(np.float32), (rs.randn(32, 16, 3, 3) / 12).astype(np.float32),
           (rs.randn(32) * 0.1).astype(np.float32), 'SAME', 1, exact=False)

    def add_ic(name, img, ksize, stride):
        cols, (oh, ow) = im2col(img, ksize, stride)
        out[name + "_img"], out[name + "_cols"] = img, cols
        ic.append({"name": name, "ksize": list(ksize), "stride": stride, "oh": int(oh), "ow": int(ow)})

    add_ic("ic_3d", rs.randn(3, 7, 9).astype(np.float32), [4, 3, 3, 3], 1)
    add_ic("ic_3d_s2", rs.randn(2, 8, 10).astype(np.float32), [1, 2, 2, 2], 2)
    add_ic("ic_4d_pool_getter", rs.randn(5, 1, 8, 6).astype(np.float32), [5, 5, 2, 2], 2)      # maxpool.py:49-50 / frame_numpy.py:54-55
    add_ic("ic_4d_batch", rs.randn(2, 3, 6, 6).astype(np.float32), [7, 3, 3, 3], 1)
    add_ic("ic_f64", rs.randn(2, 6, 6), [1, 2, 3, 3], 1)
    add_ic("ic_i32", rs.randint(0, 100, (2, 6, 7)).astype(np.int32), [1, 2, 2, 3], 1)
    out["meta"] = np.array(json.dumps({"conv2d_event": ce, "conv2d": cd, "im2col": ic, "numpy": np.__version__}))
    np.savez_compressed(os.path.join(HERE, "dense_ops.npz"), **out)
    print("dense_ops.npz: %d conv2d_event, %d conv2d, %d im2col cases" % (len(ce), len(cd), len(ic)))


def gen_feeder():
    """The step in front of the path: utils.center_crop (src/libs/utils.py:4-35), Runner.data_transform (src/libs/runner.py:23-33)
    and viz.integrate_frame (src/libs/viz.py:7-24) of the reference on seeded samples; plus the batching rule of Runner.run
    (runner.py:65-72), which is inline code of a method that needs a dataset reader and a GUI: its two NumPy expressions are
    evaluated here on the transformed events.  runner.py / viz.py import cv2 at the top (not installed, untouched by these
    functions): empty placeholder."""
    import types
    sys.path.insert(0, REF)
    _absent_module("cv2")
    from src.libs.utils import center_crop
    from src.libs.runner import Runner
    from src.libs.viz import integrate_frame
    rs = np.random.RandomState(41)
    out, cases = {}, []
    for ci, (n, eh, ew, fh, fw, xlo, xhi, ylo, yhi) in enumerate([
            (3000, 180, 240, 160, 224, 0, 240, 0, 180), (1500, 180, 240, 160, 224, 5, 236, 3, 175), (800, 34, 34, 28, 28, 0, 34, 0, 34),
            (600, 180, 240, 180, 240, 0, 240, 0, 180), (400, 64, 96, 32, 32, 10, 90, 4, 60)]):
        x, y = rs.randint(xlo, xhi, n), rs.randint(ylo, yhi, n)
        ts = np.sort(rs.randint(1000, 300000, n))
        p = rs.randint(0, 2, n)
        name = "fd%d" % ci
        out[name + "_x"], out[name + "_y"], out[name + "_ts"] = x, y, ts
        if (eh, ew) != (fh, fw):
            l2, cx, cy, cts, cp, _ = center_crop(n, x.copy(), y.copy(), ts.copy(), p.copy(), np.zeros((1, 4)), (eh, ew), (fh, fw))
            out[name + "_cx"], out[name + "_cy"], out[name + "_cts"] = cx, cy, cts
        args = types.SimpleNamespace(frame_h=fh, frame_w=fw, example_h=eh, example_w=ew)
        _, events = Runner.data_transform(n, x.copy(), y.copy(), ts.copy(), p.copy(), np.zeros((1, 4)), args)
        out[name + "_events"] = events
        for size in (200, 57):          # runner.py:71-72
            nb = int(np.ceil(events.shape[0] / size))
            out[name + "_split%d" % size] = np.cumsum([0] + [len(b) for b in np.array_split(events, indices_or_sections=nb, axis=0)])
        for usec in (10000, 777):       # runner.py:66-69
            bins = np.arange(0, events[:, -1][-1], usec)
            bin_ids = np.digitize(events[:, -1], bins)
            split_indices = np.where(bin_ids[:-1] != bin_ids[1:])[0] + 1
            out[name + "_usec%d" % usec] = np.cumsum([0] + [len(b) for b in np.array_split(events, indices_or_sections=split_indices, axis=0)])
        cases.append({"name": name, "example_hw": [eh, ew], "frame_hw": [fh, fw], "cropped": (eh, ew) != (fh, fw)})
    # integrate_frame: a sequence of batches with duplicate pixels; exact (dyadic leak) and float (leak 0.1, 5e-5)
    fcases = []
    for name, leak, h, w, nb, per in (("if_exact", 2.0 ** -6, 12, 16, 12, 9), ("if_float", 0.1, 8, 8, 20, 5), ("if_efcn", 5e-5, 160, 224, 4, 200)):
        state, prev, frames, evs = None, 0, [], []
        for t in range(nb):
            ts = np.sort(rs.randint(1, 10 if h < 100 else 1000, per)) + prev
            ev = np.stack([rs.randint(0, h, per), rs.randint(0, w, per), ts], -1)
            if t % 3 == 1:
                ev[1, :2] = ev[0, :2]; ev[-1, :2] = ev[0, :2]        # duplicate pixels inside a batch (last writer wins, viz.py:21)
            frame, prev = integrate_frame(ev, leak, h, w, state)
            state = (frame, prev)
            frames.append(frame.copy()); evs.append(ev)
        out[name + "_events"], out[name + "_frames"] = np.stack(evs).astype(np.int32), np.stack(frames)
        fcases.append({"name": name, "leak": leak, "h": h, "w": w, "dtype": str(frames[-1].dtype)})
    out["meta"] = np.array(json.dumps({"cases": cases, "integrate_frame": fcases, "numpy": np.__version__}))
    np.savez_compressed(os.path.join(HERE, "feeder.npz"), **out)
    print("feeder.npz: %d samples, %d integrate_frame traces" % (len(cases), len(fcases)))


def gen_frame_model():
    """BASELINE config 5's reference-side comparator: YoloFrameNumpy (src/models/frame_numpy.py:9-106) on frames produced by the
    reference's own integrate_frame.  The module imports tensorflow (checkpoint reader only: __init__ -> restore); neither is
    installed nor touched by `maxpool` / `cnn`: empty placeholders, and the object is created without __init__ (weights are set
    directly, transposed as build_graph does, frame_numpy.py:94-97).  Stored per case: the events, the frame, `cnn`'s output
    (activation again after every pool) and the activation-once variant (the event path's semantics,
    test_correctness.py:73-80) composed from the reference's own conv2d / maxpool / actfn."""
    import_reference()
    _absent_module("cv2")
    for m in ("tensorflow", "tensorflow.contrib", "tensorflow.contrib.framework", "tensorflow.contrib.framework.python",
              "tensorflow.contrib.framework.python.framework"):
        _absent_module(m)
    import types
    sys.modules["tensorflow.contrib.framework.python.framework"].checkpoint_utils = types.ModuleType("checkpoint_utils")
    from collections import OrderedDict
    from src.models.frame_numpy import YoloFrameNumpy
    from src.layers.conv2d import conv2d
    from src.layers.functional import make_leaky_rectified_actfn
    from src.libs.viz import integrate_frame
    out, cases = {}, []

    def run(name, spec_items, weights_hwio, h, w, leak, alpha, padding, batches, exact):
        m = object.__new__(YoloFrameNumpy)
        m._cnn_layers, m._padding, m._alpha = OrderedDict(spec_items), padding, alpha
        m._weights = {}
        for k, v in weights_hwio.items():      # frame_numpy.py:94-97
            m._weights[k] = np.ascontiguousarray(v.transpose([3, 2, 0, 1])) if k.startswith("w_conv") else v
        actfn = make_leaky_rectified_actfn(alpha)
        state, q5, once, frames = None, [], [], []
        for ev in batches:
            frame, prev = integrate_frame(ev, leak, h, w, state)
            state = (frame, prev)
            x = np.expand_dims(frame, 0)
            q5.append(np.asarray(m.cnn(x, actfn), np.float32))                      # [H, W, C]
            cur = x
            for lname, size in m._cnn_layers.items():                                # activation once
                if 'conv' in lname:
                    cur = actfn(conv2d(cur, m._weights['w_' + lname], m._weights['b_' + lname], padding))
                else:
                    cur = m.maxpool(cur, size, size[0])
            once.append(np.asarray(cur.transpose(1, 2, 0), np.float32))
            frames.append(np.asarray(frame, np.float32))
        out[name + "_events"] = np.stack(batches).astype(np.int32)
        out[name + "_frames"], out[name + "_q5"], out[name + "_once"] = np.stack(frames), np.stack(q5), np.stack(once)
        for k, v in weights_hwio.items():
            out[name + "_" + k] = v
        cases.append({"name": name, "spec": " ".join("%s=%s" % (k, ",".join(str(i) for i in v)) for k, v in spec_items), "h": h, "w": w,
                      "leak": leak, "alpha": alpha, "padding": padding, "exact": exact})

    rs = np.random.RandomState(51)
    spec = [("conv1", [3, 3, 1, 4]), ("pool1", [2, 2]), ("conv2", [3, 3, 4, 8]), ("pool2", [2, 2]), ("conv3", [1, 1, 8, 6])]
    wts = {}
    for nme, sz in spec:
        if 'conv' in nme:
            wts['w_' + nme] = rs.randint(-2, 3, size=sz).astype(np.float32)
            wts['b_' + nme] = rs.randint(-3, 4, size=sz[3]).astype(np.float32)
    batches, prev = [], 0
    for t in range(6):
        ts = np.sort(rs.randint(1, 10, 7)) + prev
        prev = int(ts.max())
        batches.append(np.stack([rs.randint(0, 16, 7), rs.randint(0, 16, 7), ts], -1))
    run("fm_exact", spec, wts, 16, 16, 2.0 ** -5, 0.125, 'SAME', batches, True)
    # the e-fcn, weights rule W (seed 1234), three batches of 200 uniform events
    layers = OrderedDict()
    for item in ("conv1=3,3,1,16 pool1=2,2 conv2=3,3,16,32 pool2=2,2 conv3=3,3,32,64 pool3=2,2 conv4=3,3,64,128 pool4=2,2 "
                 "conv5=3,3,128,256 pool5=2,2 conv6=1,1,256,512 conv7=1,1,512,110").split(' '):
        layers[item.split('=')[0]] = [int(v) for v in item.split('=')[1].split(',')]
    rw = np.random.RandomState(1234)
    wts = {}
    for nme, sz in layers.items():
        if 'conv' in nme:
            kh, kw, ci, co = sz
            wts['w_' + nme] = (rw.randn(kh, kw, ci, co) * np.sqrt(2.0 / (kh * kw * ci))).astype(np.float32)
            wts['b_' + nme] = (rw.randn(co) * 0.1).astype(np.float32)
    batches, prev = [], 0
    for t in range(3):
        ts = np.sort(rs.randint(0, 1000, 200)) + prev
        prev = int(ts.max())
        batches.append(np.stack([rs.randint(0, 160, 200), rs.randint(0, 224, 200), ts], -1))
    run("fm_efcn", list(layers.items()), wts, 160, 224, 5e-5, 0.1, 'SAME', batches, False)
    for k in list(out):                      # the e-fcn weights follow rule W: not stored (2.3 MB), regenerated from the seed
        if k.startswith("fm_efcn_w_") or k.startswith("fm_efcn_b_"):
            del out[k]
    del out["fm_efcn_frames"]                # (the frames are checked in feeder.npz; 430 KB)
    cases[-1]["weights_rule"] = {"seed": 1234}
    out["meta"] = np.array(json.dumps({"cases": cases, "numpy": np.__version__}))
    np.savez_compressed(os.path.join(HERE, "frame_model.npz"), **out)
    print("frame_model.npz: %d cases" % len(cases))


def time_reference(classes):
    """BASELINE.md section 2 protocol: e-fcn, weights rule W, uniform events, 1 warm-up step, perf_counter."""
    rs = np.random.RandomState(1234)
    layers = build_chain(classes, EFCN, he_weights(rs, EFCN), 160, 224, 5e-5, 0.1, "SAME")
    res = {}
    for B, nsteps in ((1, 30), (20, 30), (200, 50), (1000, 20), (10000, 5)):
        for l in layers:
            l.reset()
        prev, tot = 0, 0.0
        for t in range(nsteps + 1):
            ts = np.sort(rs.randint(0, max(B * 5, 2), B)) + prev
            ev = np.stack([rs.randint(0, 160, B), rs.randint(0, 224, B), ts], -1).astype(np.int32)
            prev = int(ts.max())
            t0 = time.perf_counter()
            layers[-1].compute_all(ev, None)
            out = layers[-1].featuremap()
            dt = time.perf_counter() - t0
            if t > 0:
                tot += dt
        res["B%d" % B] = {"steps": nsteps, "ms_per_step": 1e3 * tot / nsteps, "events_per_s": B * nsteps / tot}
        print("reference B=%d: %.2f ms/step, %.0f events/s" % (B, 1e3 * tot / nsteps, B * nsteps / tot))
    res["host"] = {"cpu": "Intel Xeon @ 2.10GHz (8 vCPU), one Python thread", "numpy": np.__version__}
    res["port_over_reference"] = time_port_beside_reference(layers)
    with open(os.path.join(HERE, "ref_timing.json"), "w") as f:
        json.dump(res, f, indent=1)


def time_port_beside_reference(ref_layers, B=200, nsteps=30):
    """BASELINE.md section 3 item 2 / SURVEY 8d "CPU baseline timing": the C restatement (oracle/aev_oracle.c -- a scalar fmaf chain in
    the arithmetic contract's order, what bench.py's cpu_baseline times on the GPU host) beside the reference itself (Cython ops + NumPy /
    OpenBLAS GEMMs) on IDENTICAL events, step by step alternately in this container, one thread each.  The ratio
    port events/s : reference events/s is what makes the GPU-over-port figure of the bench line traceable to the real reference."""
    sys.path.insert(0, REPO)
    from oracle import oracle as O
    from tests.util import build_chain as port_chain
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    port = port_chain(O, EFCN, he_weights(np.random.RandomState(1234), EFCN), 160, 224, 5e-5, 0.1, "SAME")
    for l in ref_layers:
        l.reset()
    rs = np.random.RandomState(4321)
    prev, t_ref, t_port = 0, 0.0, 0.0
    for t in range(nsteps + 1):
        ts = np.sort(rs.randint(0, B * 5, B)) + prev
        ev = np.stack([rs.randint(0, 160, B), rs.randint(0, 224, B), ts], -1).astype(np.int32)
        prev = int(ts.max())
        t0 = time.perf_counter()
        ref_layers[-1].compute_all(ev, None)
        ref_layers[-1].featuremap()
        t1 = time.perf_counter()
        O.net_step(port[-1], ev)
        port[-1].featuremap()
        t2 = time.perf_counter()
        if t > 0:
            t_ref += t1 - t0
            t_port += t2 - t1
    out = {"events_per_stream_step": B, "steps": nsteps, "reference_ms_per_step": 1e3 * t_ref / nsteps, "port_ms_per_step": 1e3 * t_port / nsteps,
           "reference_events_per_s": B * nsteps / t_ref, "port_events_per_s": B * nsteps / t_port, "ratio": t_ref / t_port,
           "note": "ratio = port events/s : reference events/s on identical uniform events, alternating step by step, one thread each, this container"}
    print("port beside reference (B=%d): reference %.1f ms/step, port %.1f ms/step, port : reference = %.3f" % (
        B, out["reference_ms_per_step"], out["port_ms_per_step"], out["ratio"]))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--timing", action="store_true", help="also time the reference CPU path (BASELINE.md 2)")
    ap.add_argument("--only-timing", action="store_true")
    ap.add_argument("--only-nms", action="store_true")
    ap.add_argument("--only-next-rows", action="store_true", help="only the fixtures of the rows after the path (NMS, N-format decoder, bbox decode)")
    ap.add_argument("--only-round3", action="store_true", help="only the fixtures added in round 3 (free functions, feeder, dense model)")
    ap.add_argument("--only-round4", action="store_true", help="only the fixtures added in round 4 (pool layers directly behind pool layers)")
    args = ap.parse_args()
    if args.only_round4:
        gen_pool_stacks(import_reference())
        sys.exit(0)
    if args.only_round3:
        gen_dense_ops(); gen_feeder(); gen_frame_model()
        sys.exit(0)
    if args.only_nms:
        gen_nms()
        sys.exit(0)
    if args.only_next_rows:
        gen_nms(); gen_nformat(); gen_bbox_decode()
        sys.exit(0)
    classes = import_reference()
    if not args.only_timing:
        gen_ops(classes[0], classes[1])
        gen_layers(classes)
        gen_pool_stacks(classes)
        gen_nms()
        gen_nformat()
        gen_bbox_decode()
        gen_dense_ops()
        gen_feeder()
        gen_frame_model()
    if args.timing or args.only_timing:
        time_reference(classes)
