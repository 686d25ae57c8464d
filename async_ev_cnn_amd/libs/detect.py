"""YOLO decode + NMS on the GPU -- the step right after the hot path.

Counterparts of the reference's post-processing: ``convert_bboxes`` + the decode half of ``draw_bboxes``
(src/libs/viz.py:27-46,125-164, everything before the OpenCV drawing) and ``apply_nms``
(src/libs/utils.py:38-118), computed by ``k_yolo_detect`` / ``k_nms`` through the C ABI (include/aev.h)."""
import ctypes as C

import numpy as np

from .. import _native as N


def apply_nms(batch_bboxes, batch_scores, batch_valid=None, iou_threshold=0.5):
    """Same contract as utils.py:38-118: boxes [batch, n, 4] (x_center, y_center, w, h), scores [batch, n], optional bool
    mask [batch, n]; returns [idx_axis_0, idx_axis_1] of the picked boxes, per batch entry in pick order."""
    b = np.ascontiguousarray(batch_bboxes, dtype=np.float32)
    s = np.ascontiguousarray(batch_scores, dtype=np.float32)
    if b.ndim != 3 or b.shape[-1] != 4 or s.shape != b.shape[:2]:
        raise ValueError("boxes must be [batch, n, 4] and scores [batch, n]")
    nbatch, n = s.shape
    v = None if batch_valid is None else np.ascontiguousarray(batch_valid, dtype=np.uint8)
    order, npick = np.zeros((nbatch, max(n, 1)), np.int32), np.zeros(nbatch, np.int32)
    N.check(N.lib().aev_nms(b.ctypes.data_as(N.f32p), s.ctypes.data_as(N.f32p),
                            None if v is None else v.ctypes.data_as(C.POINTER(C.c_uint8)), nbatch, n, float(iou_threshold),
                            order.ctypes.data_as(N.i32p), npick.ctypes.data_as(N.i32p), N.default_device()), "apply_nms")
    i0 = np.concatenate([np.full(k, bi, np.int64) for bi, k in enumerate(npick)]) if nbatch else np.zeros(0, np.int64)
    i1 = np.concatenate([order[bi, :k].astype(np.int64) for bi, k in enumerate(npick)]) if nbatch else np.zeros(0, np.int64)
    return [i0, i1]


def detections(net_predictions, num_classes, h_image, w_image, conf_threshold=0.2, use_nms=False, nms_threshold=0.2):
    """The decode of draw_bboxes (viz.py:125-170) for a batch of prediction tensors [S, grid_h, grid_w, C + 5B]
    (or one [grid_h, grid_w, C + 5B]).  Returns a dict of arrays over nb = grid_h*grid_w*B boxes per entry:
    bboxes [S, nb, 4] (x, y, w, h in pixels), conf [S, nb], label [S, nb] (argmax class), valid [S, nb] bool."""
    p = np.ascontiguousarray(net_predictions, dtype=np.float32)
    single = p.ndim == 3
    if single:
        p = p[None]
    s, gh, gw, d = p.shape
    nbox = (d - num_classes) // 5
    if nbox <= 0 or num_classes + 5 * nbox != d:
        raise ValueError("last dimension must be num_classes + 5 * num_bbox")
    nb = gh * gw * nbox
    out = np.empty((s, nb, 7), np.float32)
    N.check(N.lib().aev_yolo_detect(p.ctypes.data_as(N.f32p), s, gh, gw, int(num_classes), nbox, int(h_image), int(w_image),
                                    float(conf_threshold), 1 if use_nms else 0, float(nms_threshold),
                                    out.ctypes.data_as(N.f32p), None, None, N.default_device()), "detections")
    res = {"bboxes": out[..., 0:4], "conf": out[..., 4], "label": out[..., 5].astype(np.int64), "valid": out[..., 6] > 0}
    return {k: v[0] for k, v in res.items()} if single else res
