"""Conv2DLayer: asynchronous conv + leaky-ReLU (reference src/layers/conv2d.py:9-141) on the GPU."""
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
