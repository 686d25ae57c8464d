"""N-MNIST / N-Caltech101 recordings on the GPU -- the data format right before the hot path.
Counterpart of ``NReader.read_example`` (src/readers/file_reader.py:30-58): same return tuple."""
import ctypes as C

import numpy as np

from .. import _native as N


def decode(raw):
    """raw: uint8 buffer of 5-byte records -> (length, x, y, ts, p) as int32 arrays (file_reader.py:52-58)."""
    raw = np.ascontiguousarray(raw, dtype=np.uint8)
    n = len(raw) // 5
    x, y, ts, p = (np.empty(max(n, 1), np.int32) for _ in range(4))
    m = C.c_longlong(0)
    N.check(N.lib().aev_decode_nformat(raw.ctypes.data_as(C.POINTER(C.c_uint8)), len(raw), x.ctypes.data_as(N.i32p),
                                       y.ctypes.data_as(N.i32p), ts.ctypes.data_as(N.i32p), p.ctypes.data_as(N.i32p),
                                       C.byref(m), N.default_device()), "decode_nformat")
    m = m.value
    return m, x[:m].copy(), y[:m].copy(), ts[:m].copy(), p[:m].copy()


class NReader:
    """Same surface as the reference's NReader for reading (file_reader.py:24-58)."""

    def __init__(self, **kargs):
        pass

    def read_example(self, filename):
        return decode(np.fromfile(filename, dtype=np.uint8))
