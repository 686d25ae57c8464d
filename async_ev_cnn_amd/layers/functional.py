"""Host-side helpers with the names the reference's callers import (src/layers/functional.py).

`make_leaky_rectified_actfn`, `flatten` and `fully_connected` are reachable in the reference only for nets
with `fc*` layers (event_numpy.py:81-86); the e-fcn has none, they are kept as trivial NumPy utilities.
The dense `im2col` of the reference (functional.py:4-34) has no counterpart here: its two users (conv state
init, pool getters) are device kernels (k_conv_* over all sites, k_getter)."""
import numpy as np


def make_leaky_rectified_actfn(alpha):
    def actfn(v):
        return np.maximum(v, v * alpha)
    return actfn


def flatten(v, batch_axis=None):
    v = np.asarray(v)
    if batch_axis is None:
        return v.reshape(-1)
    return np.moveaxis(v, batch_axis, 0).reshape(v.shape[batch_axis], -1)


def fully_connected(v, weight, bias, activation_fn=make_leaky_rectified_actfn(0.1)):
    out = np.matmul(v, weight) + bias
    return out if activation_fn is None else activation_fn(out)
