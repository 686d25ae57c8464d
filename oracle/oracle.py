"""ctypes front-end of the CPU oracle (oracle/aev_oracle.c).

TEST INFRASTRUCTURE -- NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg may import this module; nothing under
async_ev_cnn_amd/ does.  Parity status: pinned against the golden vectors in
tests/golden/ (generated from the reference itself by tests/golden/gen_golden.py).

The classes mirror the reference's layer API (src/layers/layer.py:32-80) so that
parity tests can drive the oracle and the HIP path with the same code:
``IntegrationLayer(leak, h, w)``, ``Conv2DLayer(prev, kernel, bias, stride, alpha, padding)``,
``MaxPoolLayer(prev, ksize, stride)`` with ``reset / compute / compute_all / surface /
layer_actfn / conv_actfn / out_shape / featuremap``; plus the two ops of
src/libs/cutils.pyx (``im2col_event``, ``min_argmax``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SURFACE, LAYER_ACTFN, CONV_ACTFN, FEATUREMAP, RAW_ACT, IDX_MAX, RECOMPUTE = range(7)

_ERRORS = {-1: NotImplementedError, -2: ValueError, -3: IndexError, -4: MemoryError, -5: ValueError}


def build():
    """Compile liborc.so with gcc if it is missing or older than its source."""
    so, src = os.path.join(_HERE, "liborc.so"), os.path.join(_HERE, "aev_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        i32p, f32p, vp = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_void_p
        L.orc_im2col_event.argtypes = [f32p, C.c_int, C.c_int, C.c_int, i32p, i32p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, f32p, i32p, i32p, C.POINTER(C.c_int)]
        L.orc_min_argmax.argtypes = [f32p, f32p, C.c_int, C.c_int, i32p, i32p]
        L.orc_net_create.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(vp)]
        L.orc_net_destroy.argtypes = [vp]
        L.orc_net_destroy.restype = None
        L.orc_net_add_conv.argtypes = [vp, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_net_add_pool.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.orc_net_reset.argtypes = [vp]
        L.orc_layer_step.argtypes = [vp, C.c_int, i32p, i32p, C.c_int, C.POINTER(C.c_double)]
        L.orc_net_step.argtypes = [vp, i32p, C.c_int]
        L.orc_net_num_layers.argtypes = [vp]
        L.orc_layer_shape.argtypes = [vp, C.c_int, i32p]
        L.orc_layer_num_events.argtypes = [vp, C.c_int]
        L.orc_layer_sites_total.argtypes = [vp, C.c_int]
        L.orc_layer_sites_total.restype = C.c_longlong
        L.orc_layer_events.argtypes = [vp, C.c_int, i32p, i32p]
        L.orc_layer_read.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orc_integrate_frame.argtypes = [i32p, C.c_int, C.c_double, C.c_int, C.c_int, f32p, C.POINTER(C.c_longlong)]
        L.orc_dense_forward.argtypes = [vp, f32p, C.POINTER(f32p)]
        L.orc_dense_forward2.argtypes = [vp, f32p, C.POINTER(f32p), C.c_int]
        L.orc_conv2d_event.argtypes = [f32p, C.c_int, C.c_int, C.c_int, i32p, i32p, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, f32p, i32p, i32p, C.POINTER(C.c_int)]
        L.orc_conv2d.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, f32p]
        L.orc_im2col.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        L.orc_decode_nformat.argtypes = [C.POINTER(C.c_uint8), C.c_longlong, i32p, i32p, i32p, i32p]
        L.orc_decode_nformat.restype = C.c_longlong
        L.orc_nms.argtypes = [f32p, f32p, C.POINTER(C.c_uint8), C.c_int, C.c_float, i32p, C.POINTER(C.c_int)]
        L.orc_yolo_detect.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                                      f32p, i32p, C.POINTER(C.c_int)]
        _LIB = L
    return _LIB


def _check(rc, what):
    if rc < 0:
        raise _ERRORS.get(rc, RuntimeError)("oracle %s failed (code %d)" % (what, rc))
    return rc


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# ----------------------------------------------------------------------------- ops (src/libs/cutils.pyx)
def im2col_event(image, event_y, event_x, k_height, k_width, stride, chan_as_cols=0):
    """Same contract as cutils.pyx:29-134: returns (out_cols F-order, (out_y, out_x))."""
    if image.dtype != np.float32 or image.ndim != 3:
        raise ValueError("image must be a float32 [C,H,W] array")
    if event_y.dtype != np.int32 or event_x.dtype != np.int32 or event_y.ndim != 1 or event_x.ndim != 1:
        raise ValueError("event coordinates must be 1-D int32 arrays")
    image = np.ascontiguousarray(image)
    ch, h, w = image.shape
    if not (stride == 1 or (stride == k_height and stride == k_width)):
        raise NotImplementedError("This method only support stride equal to 1 or to the kernel's dimensions.")
    oh, ow = (h - k_height) // stride + 1, (w - k_width) // stride + 1
    kk = k_height * k_width
    rows, maxcols = (kk, ch * oh * ow) if chan_as_cols else (ch * kk, oh * ow)
    cols = np.empty((rows, maxcols), dtype=np.float32, order="F")
    oy, ox = np.empty(oh * ow, np.int32), np.empty(oh * ow, np.int32)
    n = C.c_int(0)
    ey, ex = _i32(event_y), _i32(event_x)
    _check(lib().orc_im2col_event(_p(image, C.c_float), ch, h, w, _p(ey, C.c_int32), _p(ex, C.c_int32), len(ey),
                                  k_height, k_width, stride, int(bool(chan_as_cols)), _p(cols, C.c_float),
                                  _p(oy, C.c_int32), _p(ox, C.c_int32), C.byref(n)), "im2col_event")
    n = n.value
    return cols[:, :(n * ch if chan_as_cols else n)], (oy[:n].copy(), ox[:n].copy())


def min_argmax(max_arg, min_arg):
    """Same contract as cutils.pyx:139-179."""
    for a in (max_arg, min_arg):
        if a.dtype != np.float32 or a.ndim != 2 or not a.flags.f_contiguous:
            raise ValueError("min_argmax needs float32 Fortran-ordered 2-D arrays")
    r, m = max_arg.shape
    am, na = np.empty(m, np.int32), np.empty(m, np.int32)
    _check(lib().orc_min_argmax(_p(max_arg, C.c_float), _p(min_arg, C.c_float), r, m, _p(am, C.c_int32),
                                _p(na, C.c_int32)), "min_argmax")
    return am, na


def integrate_frame(events, leak, frame_h, frame_w, prev_output):
    """viz.py:7-24 twin (float32 frame)."""
    ev = _i32(events)
    if prev_output is None:
        frame, prev_ts = np.zeros([frame_h, frame_w], np.float32), 0
    else:
        frame, prev_ts = prev_output
    new = np.array(frame, dtype=np.float32, copy=True)
    ts = C.c_longlong(int(prev_ts))
    _check(lib().orc_integrate_frame(_p(ev, C.c_int32), len(ev), float(leak), frame_h, frame_w, _p(new, C.c_float),
                                     C.byref(ts)), "integrate_frame")
    return new, ts.value


# ----------------------------------------------------------------------------- layers
class _Net:
    def __init__(self, h, w, leak):
        self.h = C.c_void_p()
        _check(lib().orc_net_create(h, w, float(leak), C.byref(self.h)), "net_create")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_net_destroy(self.h)
            self.h = None


class Layer:
    """Counterpart of src/layers/layer.py:2-80 on top of the C oracle."""

    def _setup(self, net, index):
        self._net, self._index = net, index
        chw = np.zeros(3, np.int32)
        _check(lib().orc_layer_shape(net.h, index, _p(chw, C.c_int32)), "layer_shape")
        self._out_shape = [int(v) for v in chw]

    def _read(self, which, dtype=np.float32, shape=None):
        out = np.empty(self._out_shape if shape is None else shape, dtype=dtype)
        _check(lib().orc_layer_read(self._net.h, self._index, which, out.ctypes.data_as(C.c_void_p)), "layer_read")
        return out

    def _events(self):
        n = lib().orc_layer_num_events(self._net.h, self._index)
        y, x = np.empty(n, np.int32), np.empty(n, np.int32)
        _check(lib().orc_layer_events(self._net.h, self._index, _p(y, C.c_int32), _p(x, C.c_int32)), "layer_events")
        return y, x

    def reset(self):
        # the reference resets layer by layer; callers always reset the whole chain (event_numpy.py:96-98)
        _check(lib().orc_net_reset(self._net.h), "reset")

    def surface(self):
        return self._read(SURFACE)

    def layer_actfn(self):
        return self._read(LAYER_ACTFN)

    def conv_actfn(self):
        return self._read(CONV_ACTFN)

    def featuremap(self):
        return self._read(FEATUREMAP)

    def out_shape(self):
        return self._out_shape

    def sites_total(self):
        return int(lib().orc_layer_sites_total(self._net.h, self._index))

    def compute(self, events, delta_leak):
        y, x = _i32(events[0]), _i32(events[1])
        dl = C.c_double(float(delta_leak))
        _check(lib().orc_layer_step(self._net.h, self._index, _p(y, C.c_int32), _p(x, C.c_int32), len(y),
                                    C.byref(dl)), "layer_step")
        return self._events(), delta_leak

    def compute_all(self, events, delta_leak=None):
        # conv2d.py:139-141 / maxpool.py:163-165: run every layer from the input up to this one
        layers = self._net.layers
        evs, dl = layers[0].compute(_i32(events), None)
        for l in layers[1:self._index + 1]:
            evs, dl = l.compute(evs, dl)
        return evs, dl


class IntegrationLayer(Layer):
    def __init__(self, leak, h_surface, w_surface):
        net = _Net(h_surface, w_surface, leak)
        net.layers = [self]
        self._setup(net, 0)

    def compute(self, events, _):
        ev = _i32(events)
        dl = C.c_double(0.0)
        _check(lib().orc_layer_step(self._net.h, 0, _p(ev, C.c_int32), None, len(ev), C.byref(dl)), "layer_step")
        return self._events(), np.float64(dl.value)


class Conv2DLayer(Layer):
    def __init__(self, prev_layer, kernel, bias, stride, alpha, padding="VALID"):
        if padding not in ("VALID", "SAME"):
            raise ValueError("'padding' must be either 'SAME' or 'VALID', but %s has been provided." % padding)
        net = prev_layer._net
        if prev_layer._index != len(net.layers) - 1:
            raise ValueError("oracle supports a linear chain only")
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        b = np.ascontiguousarray(bias, dtype=np.float32).reshape(-1)
        kh, kw, cin, cout = k.shape
        idx = _check(lib().orc_net_add_conv(net.h, _p(k, C.c_float), _p(b, C.c_float), kh, kw, cin, cout, int(stride),
                                            float(alpha), 1 if padding == "SAME" else 0), "add_conv")
        net.layers.append(self)
        self._setup(net, idx)

    def raw_conv_actfn(self):
        return self._read(RAW_ACT)


class MaxPoolLayer(Layer):
    def __init__(self, prev_layer, ksize, stride):
        net = prev_layer._net
        if prev_layer._index != len(net.layers) - 1:
            raise ValueError("oracle supports a linear chain only")
        idx = _check(lib().orc_net_add_pool(net.h, int(ksize[0]), int(ksize[1]), int(stride)), "add_pool")
        net.layers.append(self)
        self._setup(net, idx)

    def idx_max(self):
        c, h, w = self._out_shape
        return self._read(IDX_MAX, np.int32, [c * h * w])

    def recompute_coords(self):
        return self._read(RECOMPUTE, np.uint8, self._out_shape[1:]).astype(bool)


def net_step(last_layer, events):
    """Whole-chain step in C (no per-layer Python): used by the cpu_baseline timing."""
    ev = _i32(events)
    _check(lib().orc_net_step(last_layer._net.h, _p(ev, C.c_int32), len(ev)), "net_step")


def dense_forward(last_layer, frame, double_act=False):
    """Dense forward of the chain on ``frame`` [H,W]; returns the list of per-layer feature maps [C,H,W].
    double_act: the activation again after every pool layer, as the reference's dense NumPy model does
    (frame_numpy.py:77-78)."""
    net = last_layer._net
    f = np.ascontiguousarray(frame, dtype=np.float32)
    outs = [np.empty(l.out_shape(), np.float32) for l in net.layers]
    arr = (C.POINTER(C.c_float) * len(outs))(*[_p(o, C.c_float) for o in outs])
    _check(lib().orc_dense_forward2(net.h, _p(f, C.c_float), arr, 1 if double_act else 0), "dense_forward")
    return outs


# ----------------------------------------------------------------------------- free functions (conv2d.py:144-230, functional.py:4-34)
def _kernel_bias(kernel, bias):
    k = np.ascontiguousarray(kernel, dtype=np.float32)
    b32 = None
    if bias is not None and np.asarray(bias).dtype == np.float32:
        b32 = np.ascontiguousarray(np.asarray(bias).reshape(-1), dtype=np.float32)
    return k, b32


def conv2d_event(image, events, kernel, bias=None, padding='VALID', stride=1):
    """conv2d.py:144-181 (float32 contraction; a non-float32 bias is added in NumPy and promotes as in the reference)."""
    k, b32 = _kernel_bias(kernel, bias)
    cout, cin, kh, kw = k.shape
    ey, ex = _i32(np.asarray(events[0]).astype(np.int32)), _i32(np.asarray(events[1]).astype(np.int32))
    img = np.ascontiguousarray(np.ascontiguousarray(image).astype(np.float32))
    ch, h, w = img.shape
    if not (stride == 1 or (stride == kh and stride == kw)):
        raise NotImplementedError("This method only support stride equal to 1 or to the kernel's dimensions.")
    oh, ow = (h - kh) // stride + 1, (w - kw) // stride + 1
    buf = np.empty(cout * max(oh * ow, 1), np.float32)
    oy, ox, n = np.empty(max(oh * ow, 1), np.int32), np.empty(max(oh * ow, 1), np.int32), C.c_int(0)
    _check(lib().orc_conv2d_event(_p(img, C.c_float), ch, h, w, _p(ey, C.c_int32), _p(ex, C.c_int32), len(ey), _p(k, C.c_float),
                                  None if b32 is None else _p(b32, C.c_float), cout, kh, kw, int(stride), _p(buf, C.c_float),
                                  _p(oy, C.c_int32), _p(ox, C.c_int32), C.byref(n)), "conv2d_event")
    n = n.value
    conv = buf[:cout * n].reshape(cout, n).copy()
    if bias is not None and b32 is None:
        conv = conv + np.asarray(bias).reshape(cout, 1)
    return conv, (oy[:n].copy(), ox[:n].copy())


def conv2d(image, kernel, bias=None, padding='VALID', stride=1):
    """conv2d.py:184-230."""
    image = np.ascontiguousarray(image)
    k, b32 = _kernel_bias(kernel, bias)
    cout, cin, kh, kw = k.shape
    b, c, h, w = image.shape if image.ndim == 4 else [1, *image.shape]
    pt = pb = pl = pr = 0
    if isinstance(padding, str) and padding == 'SAME':
        if image.ndim == 4:
            raise ValueError("np.pad: three pad widths for a four-dimensional image (conv2d.py:211)")
        ph = max(kh - stride, 0) if h % stride == 0 else max(kh - (h % stride), 0)
        pw = max(kw - stride, 0) if w % stride == 0 else max(kw - (w % stride), 0)
        pt, pl = ph // 2, pw // 2
        pb, pr = ph - pt, pw - pl
    img = np.ascontiguousarray(image, dtype=np.float32)
    oh, ow = (h + pt + pb - kh) // stride + 1, (w + pl + pr - kw) // stride + 1
    out = np.empty((b, cout, oh, ow), np.float32)
    _check(lib().orc_conv2d(_p(img, C.c_float), b, c, h, w, _p(k, C.c_float), None if b32 is None else _p(b32, C.c_float), cout,
                            kh, kw, pt, pb, pl, pr, int(stride), _p(out, C.c_float)), "conv2d")
    if bias is not None and b32 is None:
        out = out + np.asarray(bias).reshape(1, cout, 1, 1)
    return out if image.ndim == 4 else out[0]


def im2col(image, ksize, stride=1):
    """functional.py:4-34."""
    image = np.ascontiguousarray(image)
    b, c, h, w = image.shape if image.ndim == 4 else [1, *image.shape]
    _, _, kh, kw = ksize
    oh, ow = (h - kh) // stride + 1, (w - kw) // stride + 1
    cols = np.empty((c * kh * kw, b * oh * ow), image.dtype)
    _check(lib().orc_im2col(image.ctypes.data, image.dtype.itemsize, b, c, h, w, int(kh), int(kw), int(stride), cols.ctypes.data),
           "im2col")
    return cols, [oh, ow]


def yolo_detect(pred, num_classes, num_bbox, h_image, w_image, conf_threshold=0.2, use_nms=False, nms_threshold=0.2):
    """viz.py:125-164 (everything before the drawing) for one prediction tensor [grid_h, grid_w, C+5B].
    Returns (out [nb,7] = x,y,w,h,conf,class,valid ; pick order int32[n_picked])."""
    p = np.ascontiguousarray(pred, dtype=np.float32)
    gh, gw, d = p.shape
    assert d == num_classes + 5 * num_bbox
    nb = gh * gw * num_bbox
    out, order, n = np.empty((nb, 7), np.float32), np.empty(nb, np.int32), C.c_int(0)
    _check(lib().orc_yolo_detect(_p(p, C.c_float), gh, gw, num_classes, num_bbox, h_image, w_image, float(conf_threshold),
                                 int(bool(use_nms)), float(nms_threshold), _p(out, C.c_float), _p(order, C.c_int32),
                                 C.byref(n)), "yolo_detect")
    return out, order[:n.value].copy()


def apply_nms(batch_bboxes, batch_scores, batch_valid=None, iou_threshold=0.5):
    """Same contract as src/libs/utils.py:38-118: returns [idx_axis_0, idx_axis_1]."""
    i0, i1 = [], []
    for b in range(len(batch_bboxes)):
        bx = np.ascontiguousarray(batch_bboxes[b], dtype=np.float32)
        sc = np.ascontiguousarray(batch_scores[b], dtype=np.float32)
        vl = None if batch_valid is None else np.ascontiguousarray(batch_valid[b], dtype=np.uint8)
        order, n = np.empty(len(bx), np.int32), C.c_int(0)
        _check(lib().orc_nms(_p(bx, C.c_float), _p(sc, C.c_float), None if vl is None else _p(vl, C.c_uint8), len(bx),
                             float(iou_threshold), _p(order, C.c_int32), C.byref(n)), "nms")
        i0 += [b] * n.value
        i1 += order[:n.value].tolist()
    return [np.array(i0), np.array(i1)]


def decode_nformat(raw):
    """NReader.read_example (file_reader.py:30-58) on a uint8 buffer: returns (length, x, y, ts, p)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = len(raw) // 5
    x, y, ts, p = (np.empty(n, np.int32) for _ in range(4))
    m = lib().orc_decode_nformat(raw.ctypes.data_as(C.POINTER(C.c_uint8)), len(raw), _p(x, C.c_int32), _p(y, C.c_int32),
                                 _p(ts, C.c_int32), _p(p, C.c_int32))
    return int(m), x[:m].copy(), y[:m].copy(), ts[:m].copy(), p[:m].copy()
