"""async_ev_cnn_amd -- MI355X (gfx950) implementation of the asynchronous event-driven conv / max-pool
forward path of marcocannici/async-ev-cnn, behind the reference's own Python layer / op API.

Python here is host plumbing only: every computation runs in hand-written HIP kernels
(csrc/) reached through the C ABI of libaev_hip.so (include/aev.h).  There is no CPU fallback:
importing works anywhere (so the ABI can be inspected), computing without a GPU raises.
"""
from . import _native  # noqa: F401
from .layers.integration import IntegrationLayer  # noqa: F401
from .layers.conv2d import Conv2DLayer  # noqa: F401
from .layers.maxpool import MaxPoolLayer  # noqa: F401
from .libs.cutils import im2col_event, min_argmax  # noqa: F401
from .libs.detect import apply_nms, detections  # noqa: F401
from .net import EventNet, YoloEventHip, layers_dict  # noqa: F401

__all__ = ["IntegrationLayer", "Conv2DLayer", "MaxPoolLayer", "im2col_event", "min_argmax", "EventNet",
           "YoloEventHip", "layers_dict"]
