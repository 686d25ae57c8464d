"""Conv2DLayer: asynchronous conv + leaky-ReLU (reference src/layers/conv2d.py:9-141) on the GPU, and the two free
functions of the same file that other callers import by name (src/models/frame_numpy.py:5): `conv2d_event`
(conv2d.py:144-181) and the dense `conv2d` (conv2d.py:184-230)."""
import ctypes as C

import numpy as np

from .. import _native as N
from .layer import Layer, _i32


class Conv2DLayer(Layer):
    def __init__(self, prev_layer, kernel, bias, stride, alpha, padding='VALID'):
        """kernel: [k_height, k_width, in_chan, out_chan]; bias: [out_chan] (conv2d.py:15-22)."""
        if padding not in N.PAD:
            raise ValueError("'padding' must be either 'SAME' or 'VALID', but %s has been provided." % padding)
        chain = prev_layer._chain
        if prev_layer is not chain.layers[-1]:
            raise NotImplementedError("only a linear chain of layers is supported")
        k = np.ascontiguousarray(kernel, dtype=np.float32)
        if k.ndim != 4:
            raise ValueError("kernel must be [k_height, k_width, in_chan, out_chan]")
        b = None if bias is None else np.ascontiguousarray(np.asarray(bias).reshape(-1), dtype=np.float32)
        kh, kw, cin, cout = k.shape
        if b is not None and b.size != cout:
            raise ValueError("bias must have out_chan entries")
        self._prev_layer, self._stride, self._alpha, self._padding = prev_layer, stride, alpha, padding
        idx = N.check(N.lib().aev_net_add_conv(chain.handle, k.ctypes.data_as(N.f32p),
                                               None if b is None else b.ctypes.data_as(N.f32p), kh, kw, cin, cout,
                                               int(stride), float(alpha), N.PAD[padding]), "aev_net_add_conv")
        self._attach(chain, idx)

    def compute(self, events, delta_leak):
        """events: (y, x) in the previous layer's output space.  Returns ((y, x) int64 row-major, delta_leak)."""
        y, x = _i32(events[0]), _i32(events[1])
        self._step(y, x, len(y), delta_leak)
        return self._last_events(np.int64), delta_leak

    def raw_conv_actfn(self):
        """The stored decay map _conv_actfn before the slope is applied (conv2d.py:62-63)."""
        return self._read(N.RAW_ACT)


def _kernel_bias(kernel, bias):
    k = np.ascontiguousarray(kernel, dtype=np.float32)       # [out_channel, in_channel, k_height, k_width]
    if k.ndim != 4:
        raise ValueError("kernel must be [out_channel, in_channel, k_height, k_width]")
    b32 = None
    if bias is not None and np.asarray(bias).dtype == np.float32:   # float32 bias: added on the device (float32 + float32)
        b32 = np.ascontiguousarray(np.asarray(bias).reshape(-1), dtype=np.float32)
        if b32.size != k.shape[0]:
            raise ValueError("bias must have out_channel entries")
    return k, b32


def _add_host_bias(conv, bias, b32, shape):
    """A bias that is not float32 (test_correctness.py:104 passes an int array, a float64 one is common) promotes the result
    exactly as `conv + bias_rows` does in the reference (conv2d.py:176-178): added on the host, in NumPy."""
    if bias is None or b32 is not None:
        return conv
    return conv + np.asarray(bias).reshape(shape)


def conv2d_event(image, events, kernel, bias=None, padding='VALID', stride=1):
    """conv2d.py:144-181.  image [in_channel, height, width] (already padded), events = (y, x), kernel
    [out_channel, in_channel, k_height, k_width].  Returns (conv [out_channel, n], (out_y int32[n], out_x int32[n])).
    The contraction runs in float32 on the GPU (kernel cast to float32; the reference's GEMM takes NumPy's promoted dtype)."""
    image = np.ascontiguousarray(image)
    k, b32 = _kernel_bias(kernel, bias)
    out_channel, in_channel, k_height, k_width = k.shape
    y_events, x_events = events
    ey = np.ascontiguousarray(np.asarray(y_events).astype(np.int32))   # conv2d.py:168-169
    ex = np.ascontiguousarray(np.asarray(x_events).astype(np.int32))
    image = np.ascontiguousarray(image.astype(np.float32))             # conv2d.py:170
    if image.ndim != 3 or image.shape[0] != in_channel:
        raise ValueError("image must be [in_channel, height, width] with the kernel's in_channel")
    if ey.ndim != 1 or ex.ndim != 1 or len(ey) != len(ex):
        raise ValueError("events must be two 1-D arrays of the same length")
    if not (stride == 1 or (stride == k_width and stride == k_height)):
        raise NotImplementedError("This method only support stride equal to 1 or to the kernel's dimensions.")
    ch, h, w = image.shape
    oh, ow = (h - k_height) // stride + 1, (w - k_width) // stride + 1
    cap = max(oh * ow, 0)
    buf = np.empty(out_channel * cap, np.float32)
    oy, ox = np.empty(cap, np.int32), np.empty(cap, np.int32)
    n = C.c_int(0)
    N.check(N.lib().aev_conv2d_event(image.ctypes.data_as(N.f32p), ch, h, w, ey.ctypes.data_as(N.i32p), ex.ctypes.data_as(N.i32p),
                                     len(ey), k.ctypes.data_as(N.f32p), None if b32 is None else b32.ctypes.data_as(N.f32p),
                                     out_channel, k_height, k_width, int(stride), buf.ctypes.data_as(N.f32p),
                                     oy.ctypes.data_as(N.i32p), ox.ctypes.data_as(N.i32p), C.byref(n), N.default_device()),
            "conv2d_event")
    n = n.value
    conv = buf[:out_channel * n].reshape(out_channel, n).copy()
    return _add_host_bias(conv, bias, b32, (out_channel, 1)), (oy[:n].copy(), ox[:n].copy())


def conv2d(image, kernel, bias=None, padding='VALID', stride=1):
    """conv2d.py:184-230.  image [batch, in_channel, height, width] or [in_channel, height, width]; padding 'SAME' pads by
    the TF rule (conv2d.py:199-213), anything else leaves the image as it is (Conv2DLayer.__init__ passes its stride in this
    slot, conv2d.py:59-60 / SURVEY Q3).  Returns [batch, out_channel, out_h, out_w] (or without the batch axis)."""
    image = np.ascontiguousarray(image)
    k, b32 = _kernel_bias(kernel, bias)
    out_channel, in_channel, k_height, k_width = k.shape
    if image.ndim not in (3, 4):
        raise ValueError("image must be [batch, in_channel, height, width] or [in_channel, height, width]")
    batch, img_channel, img_height, img_width = image.shape if image.ndim == 4 else [1, *image.shape]
    if img_channel != in_channel:
        raise ValueError("image channels do not match the kernel's in_channel")
    pt = pb = pl = pr = 0
    if isinstance(padding, str) and padding == 'SAME':
        if image.ndim == 4:   # conv2d.py:211: np.pad is given three axes -- the reference raises on a batched image
            raise ValueError("Unable to create correctly shaped tuple from ((0, 0), (pad_top, pad_bottom), (pad_left, pad_right))")
        ph = max(k_height - stride, 0) if img_height % stride == 0 else max(k_height - (img_height % stride), 0)
        pw = max(k_width - stride, 0) if img_width % stride == 0 else max(k_width - (img_width % stride), 0)
        pt, pl = ph // 2, pw // 2
        pb, pr = ph - pt, pw - pl
    img32 = np.ascontiguousarray(image, dtype=np.float32)
    oh, ow = (img_height + pt + pb - k_height) // stride + 1, (img_width + pl + pr - k_width) // stride + 1
    out = np.empty((batch, out_channel, oh, ow), np.float32)
    N.check(N.lib().aev_conv2d(img32.ctypes.data_as(N.f32p), batch, img_channel, img_height, img_width, k.ctypes.data_as(N.f32p),
                               None if b32 is None else b32.ctypes.data_as(N.f32p), out_channel, k_height, k_width, pt, pb, pl, pr,
                               int(stride), out.ctypes.data_as(N.f32p), N.default_device()), "conv2d")
    out = _add_host_bias(out, bias, b32, (1, out_channel, 1, 1))
    return out if image.ndim == 4 else out[0]
