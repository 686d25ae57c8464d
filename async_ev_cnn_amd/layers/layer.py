"""Base class of the stateful layers: the API contract of the reference's src/layers/layer.py:2-80
(reset / compute / compute_all / surface / layer_actfn / conv_actfn / out_shape / featuremap), with the
state resident on the GPU behind an aev_net handle.  Getters materialise host ndarrays on demand."""
import ctypes as C

import numpy as np

from .. import _native as N


class _Chain:
    """Owns the native handle shared by the layers of one chain (single stream: the reference's model)."""

    def __init__(self, h, w, leak, device=None):
        self.handle = C.c_void_p()
        self.device = N.default_device() if device is None else device
        N.check(N.lib().aev_net_create(int(h), int(w), float(leak), 1, self.device, C.byref(self.handle)),
                "aev_net_create")
        self.layers = []

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            N.lib().aev_net_destroy(h)
            self.handle = None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Layer:
    _chain = None
    _index = -1
    _out_shape = None

    def _attach(self, chain, index):
        self._chain, self._index = chain, index
        chw = np.zeros(3, np.int32)
        N.check(N.lib().aev_layer_shape(chain.handle, index, chw.ctypes.data_as(N.i32p)), "aev_layer_shape")
        self._out_shape = [int(v) for v in chw]
        chain.layers.append(self)

    def _read(self, which, dtype=np.float32, shape=None):
        out = np.empty(self._out_shape if shape is None else shape, dtype=dtype)
        N.check(N.lib().aev_layer_read(self._chain.handle, self._index, 0, which, out.ctypes.data_as(C.c_void_p)),
                "aev_layer_read")
        return out

    def _last_events(self, dtype):
        h = self._chain.handle
        n = C.c_int(0)
        N.check(N.lib().aev_layer_num_events(h, self._index, 0, C.byref(n)), "aev_layer_num_events")
        y, x = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        N.check(N.lib().aev_layer_events(h, self._index, 0, y.ctypes.data_as(N.i32p), x.ctypes.data_as(N.i32p),
                                         n.value, C.byref(n)), "aev_layer_events")
        return y.astype(dtype, copy=False), x.astype(dtype, copy=False)

    def _step(self, a, b, n, delta):
        d = C.c_double(0.0 if delta is None else float(delta))
        N.check(N.lib().aev_layer_step(self._chain.handle, self._index, 0, a.ctypes.data_as(N.i32p),
                                       None if b is None else b.ctypes.data_as(N.i32p), n, C.byref(d)),
                "aev_layer_step")
        return d.value

    # ---- reference API (layer.py:32-80) -------------------------------------------------------
    def reset(self):
        N.check(N.lib().aev_layer_reset(self._chain.handle, self._index, 0), "aev_layer_reset")

    def compute(self, events, delta_leak):
        raise NotImplementedError('Subclasses must override compute()')

    def compute_all(self, events, delta_leak=None):
        """Runs every layer from the input up to this one (conv2d.py:139-141, maxpool.py:163-165)."""
        layers = self._chain.layers
        evs, dl = layers[0].compute(events, None)
        for l in layers[1:self._index + 1]:
            evs, dl = l.compute(evs, dl)
        return evs, dl

    def surface(self):
        return self._read(N.SURFACE)

    def layer_actfn(self):
        return self._read(N.LAYER_ACTFN)

    def conv_actfn(self):
        return self._read(N.CONV_ACTFN)

    def featuremap(self):
        return self._read(N.FEATUREMAP)

    def out_shape(self):
        return self._out_shape
