"""Drop-in for the reference's Cython module src/libs/cutils.pyx: same names, argument meaning, return
shapes / dtypes / memory order and error behaviour; the work happens in HIP kernels (aev_im2col_event,
aev_min_argmax in include/aev.h)."""
import ctypes as C

import numpy as np

from .. import _native as N


def im2col_event(image, event_y, event_x, k_height, k_width, stride, chan_as_cols=0):
    """cutils.pyx:29-134.  Returns (out_cols float32 F-order [rows, n(*C)], (out_y int32[n], out_x int32[n]))."""
    # typed-buffer arguments of the Cython signature: wrong dtype / ndim -> ValueError (cutils.pyx:29-30)
    if not isinstance(image, np.ndarray) or image.dtype != np.float32 or image.ndim != 3:
        raise ValueError("Buffer dtype mismatch / wrong number of dimensions: image must be float32 [C,H,W]")
    for e in (event_y, event_x):
        if not isinstance(e, np.ndarray) or e.dtype != np.int32 or e.ndim != 1:
            raise ValueError("Buffer dtype mismatch / wrong number of dimensions: events must be 1-D int32")
    if len(event_y) != len(event_x):
        raise ValueError("event_y and event_x must have the same length")
    if not (stride == 1 or (stride == k_width and stride == k_height)):
        raise NotImplementedError("This method only support stride equal to 1 or to the kernel's dimensions.")
    image = np.ascontiguousarray(image)
    ch, h, w = image.shape
    oh, ow = (h - k_height) // stride + 1, (w - k_width) // stride + 1
    kk = k_height * k_width
    rows, maxcols = (kk, ch * oh * ow) if chan_as_cols else (ch * kk, oh * ow)
    cols = np.empty((rows, max(maxcols, 0)), dtype=np.float32, order='F')
    oy, ox = np.empty(max(oh * ow, 0), np.int32), np.empty(max(oh * ow, 0), np.int32)
    n = C.c_int(0)
    ey, ex = np.ascontiguousarray(event_y), np.ascontiguousarray(event_x)
    N.check(N.lib().aev_im2col_event(image.ctypes.data_as(N.f32p), ch, h, w, ey.ctypes.data_as(N.i32p),
                                     ex.ctypes.data_as(N.i32p), len(ey), int(k_height), int(k_width), int(stride),
                                     1 if chan_as_cols else 0, cols.ctypes.data_as(N.f32p), oy.ctypes.data_as(N.i32p),
                                     ox.ctypes.data_as(N.i32p), C.byref(n), N.default_device()), "im2col_event")
    n = n.value
    return cols[:, :(n * ch if chan_as_cols else n)], (oy[:n].copy(), ox[:n].copy())


def min_argmax(max_arg, min_arg):
    """cutils.pyx:139-179.  Returns (argmax int32[M], not_argmin int32[M])."""
    for a in (max_arg, min_arg):
        if not isinstance(a, np.ndarray) or a.dtype != np.float32 or a.ndim != 2 or not a.flags.f_contiguous:
            raise ValueError("ndarray is not Fortran contiguous / buffer dtype mismatch (float32 [R,M] order='F')")
    if max_arg.shape != min_arg.shape:
        raise ValueError("max_arg and min_arg must have the same shape")
    r, m = max_arg.shape
    am, na = np.empty(m, np.int32), np.empty(m, np.int32)
    N.check(N.lib().aev_min_argmax(max_arg.ctypes.data_as(N.f32p), min_arg.ctypes.data_as(N.f32p), r, m,
                                   am.ctypes.data_as(N.i32p), na.ctypes.data_as(N.i32p), N.default_device()),
            "min_argmax")
    return am, na
