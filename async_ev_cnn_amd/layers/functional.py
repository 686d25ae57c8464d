"""The free functions of the reference's src/layers/functional.py under their own names: the dense `im2col`
(functional.py:4-34) runs on the GPU (aev_im2col); `make_leaky_rectified_actfn`, `flatten` and `fully_connected` -- reachable
in the reference only for nets with `fc*` layers (event_numpy.py:81-86; the e-fcn has none) and in the dense NumPy model
(frame_numpy.py:6) -- are host NumPy one-liners, as in the reference."""
import numpy as np

from .. import _native as N


def im2col(image, ksize, stride=1):
    """functional.py:4-34.  image [batch, in_channel, height, width] or [in_channel, height, width]; ksize
    [out_channel, in_channel, k_height, k_width].  Returns (img_cols [img_channel*k_height*k_width, batch*out_h*out_w] of the
    image's dtype, [out_h, out_w])."""
    image = np.asarray(image)
    if image.ndim not in (3, 4):
        raise ValueError("image must be [batch, in_channel, height, width] or [in_channel, height, width]")
    if image.dtype.itemsize not in (4, 8):
        raise ValueError("im2col on the GPU copies 4- or 8-byte elements (float32 / int32 / float64 / int64)")
    batch, img_channel, img_height, img_width = image.shape if image.ndim == 4 else [1, *image.shape]
    _, _, k_height, k_width = ksize
    out_h = (img_height - k_height) // stride + 1
    out_w = (img_width - k_width) // stride + 1
    src = np.ascontiguousarray(image)
    cols = np.empty((img_channel * k_height * k_width, batch * out_h * out_w), image.dtype)
    N.check(N.lib().aev_im2col(src.ctypes.data, image.dtype.itemsize, batch, img_channel, img_height, img_width, int(k_height),
                               int(k_width), int(stride), cols.ctypes.data, N.default_device()), "im2col")
    return cols, [out_h, out_w]


def make_leaky_rectified_actfn(alpha):
    """functional.py:37-47."""
    def actfn(input):
        return np.maximum(input, input * alpha)
    return actfn


def flatten(input, batch_axis=None):
    """functional.py:50-70: flattens while keeping the batch axis (moved to the front)."""
    in_shape = list(input.shape)
    if batch_axis is not None:
        batch_size = in_shape.pop(batch_axis)
        axes = [batch_axis] + [a for a in range(input.ndim) if a != batch_axis]
        return np.reshape(input.transpose(axes), [batch_size, int(np.prod(in_shape))])
    return np.reshape(input, [-1])


def fully_connected(input, weight, bias, activation_fn=make_leaky_rectified_actfn(0.1)):
    """functional.py:73-90."""
    out = np.matmul(input, weight) + bias
    return out if activation_fn is None else activation_fn(out)
