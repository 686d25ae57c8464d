"""CPU: pin the oracle (oracle/aev_oracle.c) against golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  This is what makes the oracle a trustworthy checker for the HIP path."""
import json

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import GOLDEN, replay_against_golden

OPS = np.load(GOLDEN + "/ops.npz")
OPS_META = json.loads(str(OPS["meta"]))


@pytest.mark.parametrize("case", OPS_META["im2col"], ids=lambda c: c["name"])
def test_im2col_event_matches_reference(case):
    n = case["name"]
    cols, (oy, ox) = O.im2col_event(OPS[n + "_img"], OPS[n + "_ey"], OPS[n + "_ex"], case["kh"], case["kw"],
                                    case["stride"], case["chan_as_cols"])
    ref = OPS[n + "_cols"]
    assert cols.shape == ref.shape and cols.flags.f_contiguous or cols.size == 0
    assert np.array_equal(cols, ref)                      # pure copy: bit-exact (cutils.pyx:127)
    assert oy.dtype == np.int32 and np.array_equal(oy, OPS[n + "_oy"])   # first-touch order (cutils.pyx:108-112)
    assert np.array_equal(ox, OPS[n + "_ox"])


@pytest.mark.parametrize("name", OPS_META["min_argmax"])
def test_min_argmax_matches_reference(name):
    am, na = O.min_argmax(np.asfortranarray(OPS[name + "_mx"]), np.asfortranarray(OPS[name + "_mn"]))
    assert np.array_equal(am, OPS[name + "_am"]) and np.array_equal(na, OPS[name + "_na"])


def test_op_errors_match_reference():
    img = np.zeros((1, 6, 6), np.float32)
    e = np.zeros(1, np.int32)
    with pytest.raises(NotImplementedError):       # cutils.pyx:89
        O.im2col_event(img, e, e, 3, 3, 2)
    with pytest.raises(ValueError):                # Cython typed-buffer mismatch
        O.im2col_event(img.astype(np.float64), e, e, 3, 3, 1)
    with pytest.raises(ValueError):
        O.min_argmax(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32))  # C-order, not Fortran
    with pytest.raises(ValueError):                # conv2d.py:56
        O.Conv2DLayer(O.IntegrationLayer(0.1, 8, 8), np.zeros((3, 3, 1, 1), np.float32), np.zeros(1), 1, 0.1, "FULL")


EXACT = ["exact_testnet8", "exact_cfg1_c1_e1", "exact_cfg1_c1_e5", "exact_cfg1_c16_e1", "exact_cfg1_c16_e5",
         "exact_net32", "exact_net32_reset", "exact_net34_valid", "exact_poolstack32", "exact_poolstack32_reset", "exact_poolstack_triple"]
FLOAT = ["float_testnet8", "float_cfg1_c16_e5", "float_net32", "float_efcn3", "float_poolstack32"]


@pytest.mark.parametrize("name", EXACT)
def test_layers_exact_family(name):
    replay_against_golden(O, name)


@pytest.mark.parametrize("name", FLOAT)
def test_layers_float_family(name):
    replay_against_golden(O, name, atol=1e-5)


def test_reset_restores_construction_state():
    """reset(): state after reset equals state at construction (conv2d.py:99-103, maxpool.py:84-90)."""
    from tests.util import Golden, chain_from_golden, step_layers
    g = Golden("exact_net32")
    layers = chain_from_golden(O, g)
    init = [(l.featuremap(), l.conv_actfn()) for l in layers]
    for t in range(10):
        step_layers(layers, g.step_events(t))
    for l in layers:
        l.reset()
    for l, (fm, act) in zip(layers, init):
        assert np.array_equal(l.featuremap(), fm) and np.array_equal(l.conv_actfn(), act)


def test_nms_matches_reference():
    """NEXT ROW: the oracle's NMS against the reference's own apply_nms (src/libs/utils.py:38-118)."""
    z = np.load(GOLDEN + "/nms.npz")
    for case in json.loads(str(z["meta"]))["cases"]:
        n = case["name"]
        valid = z[n + "_v"]
        i0, i1 = O.apply_nms(z[n + "_b"], z[n + "_s"], valid, iou_threshold=case["thr"])
        assert np.array_equal(i0, z[n + "_i0"]) and np.array_equal(i1, z[n + "_i1"]), n


def _encode_nformat(x, y, ts, p, overflow_every=None):
    """Test-side encoder of the 5-byte event format (x, y, p<<7|ts[22:16], ts[15:8], ts[7:0]); inserts y == 240 overflow
    markers so that 23-bit timestamps wrap the way the recordings do."""
    out, carry = [], 0
    for i in range(len(x)):
        while ts[i] - carry >= (1 << 23) or (overflow_every and i and i % overflow_every == 0 and ts[i] - carry >= (1 << 13)):
            out.append([0, 240, 0, 0, 0])
            carry += 1 << 13
        t = int(ts[i] - carry)
        out.append([int(x[i]), int(y[i]), (int(p[i]) << 7) | ((t >> 16) & 127), (t >> 8) & 255, t & 255])
    return np.array(out, np.uint8).reshape(-1)


def _decode_numpy(raw):
    """Independent transcription of the documented slicing of file_reader.py:36-56 (not the oracle's loop)."""
    r = raw[:len(raw) // 5 * 5].astype(np.uint32)
    ax, ay = r[0::5], r[1::5]
    ap = (r[2::5] & 128) >> 7
    at = ((r[2::5] & 127) << 16) | (r[3::5] << 8) | r[4::5]
    at = at + (np.cumsum(ay == 240) * (1 << 13)).astype(np.uint32)
    td = ay != 240
    return int(td.sum()), ax[td].astype(np.int32), ay[td].astype(np.int32), at[td].astype(np.int32), ap[td].astype(np.int32)


def test_nformat_decoder_kat_and_roundtrip():
    """NEXT ROW, parity unpinned (see oracle header): hand-assembled bytes + independent NumPy transcription."""
    raw = np.array([10, 20, 0x80 | 0x01, 0x02, 0x03,      # x=10 y=20 p=1 ts=0x010203
                    0, 240, 0, 0, 0,                       # overflow marker: +8192 from here on
                    200, 100, 0x00, 0x00, 0x05], np.uint8) # x=200 y=100 p=0 ts=5+8192
    n, x, y, ts, p = O.decode_nformat(raw)
    assert n == 2 and list(x) == [10, 200] and list(y) == [20, 100] and list(p) == [1, 0] and list(ts) == [0x010203, 5 + 8192]
    rs = np.random.RandomState(0)
    m = 20000
    xs, ys, ps = rs.randint(0, 240, m), rs.randint(0, 180, m), rs.randint(0, 2, m)
    tss = np.cumsum(rs.randint(0, 900, m))
    raw = _encode_nformat(xs, ys, tss, ps, overflow_every=997)
    got = O.decode_nformat(raw)
    ref = _decode_numpy(raw)
    assert got[0] == ref[0] == m
    for a, b in zip(got[1:], ref[1:]):
        assert np.array_equal(a, b)
    assert np.array_equal(got[1], xs) and np.array_equal(got[2], ys) and np.array_equal(got[3], tss) and np.array_equal(got[4], ps)


def test_nformat_decoder_matches_reference():
    """NEXT ROW, PINNED: the oracle's decoder against NReader.read_example of the reference itself
    (src/readers/file_reader.py:30-58; tests/golden/gen_golden.py: gen_nformat) -- plain streams, overflow markers (first and
    last record included), a single event, markers only."""
    z = np.load(GOLDEN + "/nformat.npz")
    for case in json.loads(str(z["meta"]))["cases"]:
        n = case["name"]
        length, x, y, ts, p = O.decode_nformat(z[n + "_raw"])
        assert length == case["length"], n
        for got, key in ((x, "_x"), (y, "_y"), (ts, "_ts"), (p, "_p")):
            assert got.dtype == z[n + key].dtype == np.int32 and np.array_equal(got, z[n + key]), (n, key)


def test_bbox_decode_matches_reference():
    """NEXT ROW, PINNED: the box geometry of the oracle's YOLO decode against convert_bboxes of the reference
    itself (src/libs/viz.py:27-46 with draw_bboxes' arguments, sqrt=True; gen_golden.py: gen_bbox_decode) -- bit for bit
    (float32 add, divide, multiply in the reference's order)."""
    z = np.load(GOLDEN + "/bbox_decode.npz")
    for case in json.loads(str(z["meta"]))["cases"]:
        n = case["name"]
        out, _ = O.yolo_detect(z[n + "_pred"], case["C"], case["B"], case["h_image"], case["w_image"], -1e30, False, 0.0)
        assert z[n + "_boxes"].dtype == np.float32 and np.array_equal(out[:, :4], z[n + "_boxes"]), n


# ------------------------------------------------------------------------------------------------ round 3 fixtures
DOPS = np.load(GOLDEN + "/dense_ops.npz")
DOPS_META = json.loads(str(DOPS["meta"]))
FEED = np.load(GOLDEN + "/feeder.npz")
FEED_META = json.loads(str(FEED["meta"]))
FMOD = np.load(GOLDEN + "/frame_model.npz")
FMOD_META = json.loads(str(FMOD["meta"]))


def check_conv2d_event(backend, case, z=DOPS):
    n = case["name"]
    conv, (oy, ox) = backend.conv2d_event(z[n + "_img"], (z[n + "_ey"], z[n + "_ex"]), z[n + "_k"],
                                          z[n + "_b"] if case["bias"] else None, stride=case["stride"])
    ref = z[n + "_conv"]
    assert conv.shape == ref.shape and str(conv.dtype) == case["dtype"], (n, conv.shape, conv.dtype)
    assert oy.dtype == np.int32 and np.array_equal(oy, z[n + "_oy"]) and np.array_equal(ox, z[n + "_ox"]), n
    if case["exact"]:
        assert np.array_equal(conv, ref), n
    else:
        assert np.allclose(conv, ref, rtol=1e-5, atol=1e-5), (n, np.abs(conv - ref).max())
    return conv


def check_conv2d(backend, case, z=DOPS):
    n = case["name"]
    out = backend.conv2d(z[n + "_img"], z[n + "_k"], z[n + "_b"] if case["bias"] else None, case["padding"], case["stride"])
    ref = z[n + "_out"]
    assert out.shape == ref.shape and str(out.dtype) == case["dtype"], (n, out.shape, out.dtype)
    if case["exact"]:
        assert np.array_equal(out, ref), n
    else:
        assert np.allclose(out, ref, rtol=1e-5, atol=1e-5), (n, np.abs(out - ref).max())
    return out


def check_im2col(backend, case, z=DOPS):
    n = case["name"]
    cols, (oh, ow) = backend.im2col(z[n + "_img"], case["ksize"], case["stride"])
    ref = z[n + "_cols"]
    assert (oh, ow) == (case["oh"], case["ow"]) and cols.shape == ref.shape and cols.dtype == ref.dtype, n
    assert cols.flags.c_contiguous and np.array_equal(cols, ref), n


@pytest.mark.parametrize("case", DOPS_META["conv2d_event"], ids=lambda c: c["name"])
def test_conv2d_event_matches_reference(case):
    check_conv2d_event(O, case)


@pytest.mark.parametrize("case", DOPS_META["conv2d"], ids=lambda c: c["name"])
def test_conv2d_matches_reference(case):
    check_conv2d(O, case)


@pytest.mark.parametrize("case", DOPS_META["im2col"], ids=lambda c: c["name"])
def test_dense_im2col_matches_reference(case):
    check_im2col(O, case)


def test_conv2d_same_padding_on_a_batch_raises_like_the_reference():
    with pytest.raises(ValueError):     # conv2d.py:211: np.pad gets three axes for a four-dimensional image
        O.conv2d(np.zeros((2, 1, 6, 6), np.float32), np.zeros((1, 1, 3, 3), np.float32), None, 'SAME', 1)


@pytest.mark.parametrize("case", FEED_META["integrate_frame"], ids=lambda c: c["name"])
def test_integrate_frame_matches_reference(case):
    """viz.integrate_frame (viz.py:7-24), the dense twin of the integration layer: bit for bit on the exact trace; within
    1e-6 where the reference's float64 scalar meets its float32 frame (NumPy >= 2 computes that subtraction in float64)."""
    n = case["name"]
    state = None
    for t, ev in enumerate(FEED[n + "_events"]):
        frame, prev = O.integrate_frame(ev, case["leak"], case["h"], case["w"], state)
        state = (frame, prev)
        ref = FEED[n + "_frames"][t]
        assert prev == int(ev[:, 2].max())
        if n == "if_exact":
            assert np.array_equal(frame, ref), (n, t)
        else:
            assert np.allclose(frame, ref, rtol=0, atol=1e-6), (n, t, np.abs(frame - ref).max())


@pytest.mark.parametrize("case", FMOD_META["cases"], ids=lambda c: c["name"])
def test_dense_forward_matches_reference_frame_model(case):
    """orc_dense_forward against the reference's dense NumPy model (frame_numpy.py:62-88) on the reference's own frames:
    `cnn` as written (activation again after every pool) and the activation-once composition."""
    from tests.util import build_chain, he_weights
    n = case["name"]
    names = [it.split("=")[0] for it in case["spec"].split(" ")]
    sizes = [[int(v) for v in it.split("=")[1].split(",")] for it in case["spec"].split(" ")]
    spec = [("conv", *sz) if "conv" in nm else ("pool", sz[0]) for nm, sz in zip(names, sizes)]
    if case.get("weights_rule"):
        weights = he_weights(np.random.RandomState(case["weights_rule"]["seed"]), spec)
    else:
        weights = [(FMOD["%s_w_%s" % (n, nm)], FMOD["%s_b_%s" % (n, nm)]) for nm in names if "conv" in nm]
    chain = build_chain(O, spec, weights, case["h"], case["w"], case["leak"], case["alpha"], case["padding"])
    state = None
    for t, ev in enumerate(FMOD[n + "_events"]):
        frame, prev = O.integrate_frame(ev, case["leak"], case["h"], case["w"], state)
        state = (frame, prev)
        for key, dbl in (("_once", False), ("_q5", True)):
            got = O.dense_forward(chain[-1], frame, double_act=dbl)[-1].transpose(1, 2, 0)
            ref = FMOD[n + key][t]
            if case["exact"]:
                assert np.array_equal(got, ref), (n, t, key)
            else:
                assert np.allclose(got, ref, rtol=1e-5, atol=1e-5), (n, t, key, np.abs(got - ref).max())
    assert not np.array_equal(FMOD[n + "_q5"], FMOD[n + "_once"])      # the quirk is visible on this input
