"""IntegrationLayer: the leaky event surface (reference src/layers/integration.py:6-95) on the GPU."""
import numpy as np

from .layer import Layer, _Chain, _i32


class IntegrationLayer(Layer):
    def __init__(self, leak, h_surface, w_surface, device=None):
        self._leak = leak
        self._attach(_Chain(h_surface, w_surface, leak, device), 0)

    def compute(self, events, _):
        """events: int [N,3] rows (y, x, ts) (integration.py:56).  Returns ((y, x) int64 row-major, delta_leak)."""
        ev = _i32(events)
        if ev.ndim != 2 or ev.shape[1] != 3:
            raise ValueError("events must be an [N,3] array of (y, x, ts)")
        dl = self._step(ev, None, len(ev), None)
        return self._last_events(np.int64), np.float64(dl)

    def compute_all(self, events, delta_leak=None):
        return self.compute(events, None)
