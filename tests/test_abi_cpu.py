"""CPU: the C-ABI library builds, loads and exports every symbol include/aev.h declares; the product path
fails loudly without a GPU (no CPU fallback); host-side logic (spec grammar, weights rule, feeder, byte model)."""
import os
import re

import numpy as np
import pytest

import async_ev_cnn_amd as A
from async_ev_cnn_amd import _native as N
from async_ev_cnn_amd import streams as S

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return N.device_count() > 0


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "aev.h")).read()
    declared = set(re.findall(r"\b(aev_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 28
    assert declared == set(N.PROTOTYPES), declared ^ set(N.PROTOTYPES)
    lib = N.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.aev_version() >= 100
    assert lib.aev_kernel_name(2).decode() == "k_leak_sweep"


def test_product_path_never_touches_the_oracle():
    """Nothing under async_ev_cnn_amd/ may import or link anything under oracle/."""
    pkg = os.path.join(REPO, "async_ev_cnn_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and False, os.path.join(root, f)


def test_nothing_built_from_the_reference_travels_to_the_gpu_box():
    """SURVEY 8c, "GPU-box rule": no file built from the reference (the Cython module oracle/build_ref.sh compiles from cutils.pyx) may lie
    in the tree gpurun pushes -- it must sit under a path that is BOTH git-ignored and gpurun-ignored (oracle/_ref/)."""
    def patterns(name):
        return [ln.strip() for ln in open(os.path.join(REPO, name)) if ln.strip() and not ln.startswith("#")]
    gpu_ign, git_ign = patterns(".gpurunignore"), patterns(".gitignore")
    assert "oracle/_ref/" in gpu_ign and "oracle/_ref/" in git_ign
    for root, dirs, files in os.walk(REPO):
        rel = os.path.relpath(root, REPO)
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out")]
        for f in files:
            if re.match(r"cutils.*\.(so|c|pyd)$", f):
                where = os.path.join(rel, f)
                assert where.startswith("oracle/_ref/"), "a build product of the reference outside oracle/_ref/: " + where


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    with pytest.raises(RuntimeError):
        A.IntegrationLayer(0.1, 8, 8)
    with pytest.raises(RuntimeError):
        A.im2col_event(np.zeros((1, 6, 6), np.float32), np.zeros(1, np.int32), np.zeros(1, np.int32), 3, 3, 1)


def test_handle_state_is_never_initialised_on_the_null_stream():
    """Regression guard of the race round 1 shipped (HIP runtime): hipMemset runs on the NULL stream, which a handle's
    non-blocking stream does not wait for.  Every initialisation of handle state is a hipMemsetAsync on the handle's own stream:
    the host runtime contains no synchronous hipMemset at all, and the kernels launched on stream 0 are the ones of the
    stand-alone host-pointer ops (which own their buffers and synchronise through their blocking copies)."""
    src = open(os.path.join(REPO, "async_ev_cnn_amd", "csrc", "aev_hip.hip")).read()
    assert not re.findall(r"\bhipMemset\s*\(", src)
    # stream-0 launches only inside the one-shot ops (between their `int aev_<op>(` and the next top-level function)
    ops = ("aev_im2col_event", "aev_min_argmax", "aev_conv2d_event", "aev_conv2d", "aev_im2col", "aev_yolo_detect", "aev_nms", "aev_decode_nformat")
    for m in re.finditer(r"hipLaunchKernelGGL\([^;]*?, 0, 0, ", src):
        head = src[:m.start()]
        last_fn = re.findall(r"\n(?:static )?int (\w+)\(", head)[-1]
        assert last_fn in ops, "a kernel is launched on the NULL stream inside %s" % last_fn


def test_argument_validation_happens_before_the_device():
    img = np.zeros((1, 6, 6), np.float32)
    e = np.zeros(1, np.int32)
    with pytest.raises(NotImplementedError):          # cutils.pyx:89
        A.im2col_event(img, e, e, 3, 3, 2)
    with pytest.raises(ValueError):                   # typed buffer: wrong dtype
        A.im2col_event(img.astype(np.float64), e, e, 3, 3, 1)
    with pytest.raises(ValueError):
        A.im2col_event(img, e.astype(np.int64), e, 3, 3, 1)
    with pytest.raises(ValueError):                   # C order where Fortran order is required
        A.min_argmax(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32))


def test_layer_spec_grammar():
    d = A.layers_dict(A.net.EFCN_LAYERS)              # configs/efcn_event.yml:15 through config.py:6-12
    assert list(d)[:3] == ["conv1", "pool1", "conv2"] and d["conv1"] == [3, 3, 1, 16] and d["pool5"] == [2, 2]
    assert d["conv7"] == [1, 1, 512, 110] and len(d) == 12
    with pytest.raises(ValueError):
        A.layers_dict("conv1=3,3,a")


def test_weights_rule_is_frozen():
    w = A.net.he_init_weights(A.layers_dict(A.net.EFCN_LAYERS), 1234)
    assert w["w_conv1"].shape == (3, 3, 1, 16) and w["w_conv1"].dtype == np.float32
    from tests.util import EFCN, he_weights
    ref = he_weights(np.random.RandomState(1234), EFCN)
    assert np.array_equal(w["w_conv1"], ref[0][0]) and np.array_equal(w["b_conv7"], ref[6][1])


def test_center_crop_matches_reference_semantics():
    # utils.py:17-26 on a full 180x240 sensor: x window starts at (179-160)//2 = 9, y window at (239-224)//2 = 7
    yy, xx = np.mgrid[0:180, 0:240]
    x, y, ts = xx.ravel().copy(), yy.ravel().copy(), np.arange(xx.size)
    nx, ny, nts = S.center_crop_events(x, y, ts, (160, 224))
    assert nx.min() == 0 and ny.min() == 0 and nx.max() == 223 and ny.max() == 159 and len(nx) == 160 * 224
    keep = (x >= 9) & (x < 233) & (y >= 7) & (y < 167)
    assert np.array_equal(nts, ts[keep])


def test_batching_rule():
    x, y, ts = S.ncaltech_like_sample(1234, n_events=5000)
    b = S.sample_to_batches(x, y, ts, (160, 224), example_hw=(180, 240), batch_event_size=200)
    n = sum(len(v) for v in b)
    assert len(b) == int(np.ceil(n / 200.0)) and max(len(v) for v in b) - min(len(v) for v in b) <= 1  # np.array_split
    assert b[0][0, 2] >= 0 and all(v.dtype == np.int32 and v.shape[1] == 3 for v in b)
    assert all((v[:, 0] < 160).all() and (v[:, 1] < 224).all() for v in b)
    tb = S.sample_to_batches(x, y, ts, (160, 224), example_hw=(180, 240), batch_event_usec=10000)
    assert sum(len(v) for v in tb) == n
    u = S.uniform_batches(7, (160, 224), 3, 200, 200)
    assert len(u) == 3 and u[1][:, 2].min() > u[0][:, 2].max()


def test_feeder_matches_reference_fixtures():
    """center_crop (utils.py:4-35), Runner.data_transform (runner.py:23-33) and the batching rule (runner.py:65-72) against what
    the reference's own functions returned on seeded samples (tests/golden/feeder.npz)."""
    import json
    from tests.util import GOLDEN
    z = np.load(GOLDEN + "/feeder.npz")
    for case in json.loads(str(z["meta"]))["cases"]:
        n = case["name"]
        x, y, ts = z[n + "_x"], z[n + "_y"], z[n + "_ts"]
        if case["cropped"]:
            cx, cy, cts = S.center_crop_events(x, y, ts, case["frame_hw"])
            assert np.array_equal(cx, z[n + "_cx"]) and np.array_equal(cy, z[n + "_cy"]) and np.array_equal(cts, z[n + "_cts"]), n
        ev = z[n + "_events"]
        for size in (200, 57):
            b = S.sample_to_batches(x, y, ts, case["frame_hw"], case["example_hw"], batch_event_size=size)
            assert np.array_equal(np.concatenate(b), ev), n
            assert np.array_equal(np.cumsum([0] + [len(v) for v in b]), z[n + "_split%d" % size]), (n, size)
        for usec in (10000, 777):
            b = S.sample_to_batches(x, y, ts, case["frame_hw"], case["example_hw"], batch_event_usec=usec)
            assert np.array_equal(np.concatenate(b), ev), n
            assert np.array_equal(np.cumsum([0] + [len(v) for v in b]), z[n + "_usec%d" % usec]), (n, usec)


def test_shard_streams():
    assert S.shard_streams(8, 3, 8) == [3]
    assert sorted(sum((S.shard_streams(10, r, 4) for r in range(4)), [])) == list(range(10))


def test_algorithmic_byte_model_matches_survey_example():
    """SURVEY.md 8d worked example: B=200 steady-state site counts -> ~47.4 MB and ~685 MFLOP per step."""
    class Fake(A.EventNet):
        def __init__(self):
            layers = A.layers_dict(A.net.EFCN_LAYERS)
            self.h, self.w = 160, 224
            self.names = ['intgr'] + list(layers)
            self.kinds = ['intgr'] + ['conv' if 'conv' in n else 'pool' for n in layers]
            self.conv_specs = [None] + [tuple(layers[n]) for n in layers]
            hw = [(160, 224), (160, 224), (80, 112), (80, 112), (40, 56), (40, 56), (20, 28), (20, 28), (10, 14), (10, 14), (5, 7), (5, 7), (5, 7)]
            ch = [1, 16, 16, 32, 32, 64, 64, 128, 128, 256, 256, 512, 110]
            self.shapes = [[c, h, w] for c, (h, w) in zip(ch, hw)]
            self.handle = None
    sites = [306, 6193, 6693, 8659, 2204, 2238, 560, 560, 140, 140, 35, 35, 35]
    by, fl, dense = Fake().algorithmic_bytes_flops(sites, 200, 1)
    assert abs(dense - 13.88e6) < 0.02e6
    assert abs(by - 47.4e6) < 0.3e6
    assert abs(fl - 685e6) < 5e6
