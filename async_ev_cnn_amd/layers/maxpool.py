"""MaxPoolLayer: asynchronous max-pool with persistent argmax (reference src/layers/maxpool.py:8-165) on the GPU."""
import numpy as np

from .. import _native as N
from .layer import Layer, _i32


class MaxPoolLayer(Layer):
    def __init__(self, prev_layer, ksize, stride):
        chain = prev_layer._chain
        if prev_layer is not chain.layers[-1]:
            raise NotImplementedError("only a linear chain of layers is supported")
        self._prev_layer, self._ksize, self._stride = prev_layer, list(ksize), stride
        idx = N.check(N.lib().aev_net_add_pool(chain.handle, int(ksize[0]), int(ksize[1]), int(stride)),
                      "aev_net_add_pool")
        self._attach(chain, idx)

    def compute(self, events, delta_leak):
        """Returns ([out_y, out_x] int32 in first-touch order (maxpool.py:154), delta_leak)."""
        y, x = _i32(events[0]), _i32(events[1])
        self._step(y, x, len(y), delta_leak)
        oy, ox = self._last_events(np.int32)
        return [oy, ox], delta_leak

    def idx_max(self):
        """_idx_max[0]: int32 [C*H*W] window-relative argmax (maxpool.py:33-35)."""
        c, h, w = self._out_shape
        return self._read(N.IDX_MAX, np.int32, [c * h * w])

    def recompute_coords(self):
        """_recompute_coords: bool [H,W] (maxpool.py:36)."""
        return self._read(N.RECOMPUTE, np.uint8, self._out_shape[1:]).astype(bool)
