"""Model assembly and the multi-stream engine.

* ``layers_dict``   -- the layer-spec mini-grammar of the reference's config (src/scripts/config.py:6-12).
* ``YoloEventHip``  -- counterpart of ``YoloEventNumpy`` (src/models/event_numpy.py:12-105): builds the layer
                       chain from the ordered spec and returns ``graph(events, reset)``.  Weights come from a
                       dict (``w_<name>`` / ``b_<name>``) instead of a TF checkpoint (checkpoint loading is
                       out of scope).
* ``EventNet``      -- the same chain replicated over S independent event streams on one GPU, all streams
                       advanced by one launch sequence per step; device-resident event queue, hipGraph replay,
                       per-kernel timing.  This is what bench.py drives.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from . import _native as N


def layers_dict(text):
    """'conv1=3,3,1,16 pool1=2,2 ...' -> OrderedDict(name -> [ints])  (config.py:6-12)."""
    try:
        return OrderedDict((item.split('=')[0], [int(v) for v in item.split('=')[1].split(',')])
                           for item in text.split(' ') if item)
    except Exception:
        raise ValueError("Format must be 'name1=h1,w1,i1,o1 name2=h2,w2 ...'")


EFCN_LAYERS = ("conv1=3,3,1,16 pool1=2,2 conv2=3,3,16,32 pool2=2,2 conv3=3,3,32,64 pool3=2,2 conv4=3,3,64,128 "
               "pool4=2,2 conv5=3,3,128,256 pool5=2,2 conv6=1,1,256,512 conv7=1,1,512,110")  # configs/efcn_event.yml:15


def he_init_weights(cnn_layers, seed=1234):
    """Random-init weights, rule W of SURVEY.md 8d: per conv in layer order from RandomState(seed),
    w = randn(k,k,cin,cout) * sqrt(2/(k*k*cin)), b = randn(cout) * 0.1, float32."""
    rs = np.random.RandomState(seed)
    weights = {}
    for name, size in cnn_layers.items():
        if 'conv' in name:
            kh, kw, ci, co = size
            weights['w_' + name] = (rs.randn(kh, kw, ci, co) * np.sqrt(2.0 / (kh * kw * ci))).astype(np.float32)
            weights['b_' + name] = (rs.randn(co) * 0.1).astype(np.float32)
    return weights


class EventNet:
    """One layer chain x S streams on one GPU (include/aev.h).  Layer kinds are chosen by substring of the
    spec name exactly as event_numpy.py:61-71 does ('conv' -> async conv stride 1, 'pool' -> pool with
    stride = size[0])."""

    def __init__(self, h_frame, w_frame, cnn_layers, weights, padding='SAME', alpha=0.1, leak=5e-5, n_streams=1,
                 device=None, tuning="auto", conv_math="f32"):
        if isinstance(cnn_layers, str):
            cnn_layers = layers_dict(cnn_layers)
        self.h, self.w, self.n_streams = int(h_frame), int(w_frame), int(n_streams)
        self.device = N.default_device() if device is None else int(device)
        self.handle = C.c_void_p()
        N.check(N.lib().aev_net_create(self.h, self.w, float(leak), self.n_streams, self.device, C.byref(self.handle)),
                "aev_net_create")
        N.check(N.lib().aev_net_set_tuning(self.handle, {"auto": 0, "throughput": 1, "latency": 2}[tuning]), "aev_net_set_tuning")
        # "f32": the fmaf chain of the arithmetic contract (bit-identical to the CPU restatement the tests check against); "bf16x6": three-way split operands on the bf16
        # MFMA (include/aev.h: aev_net_set_conv_math) -- as accurate, 2.4x the MFMA rate, not bit-identical
        self.conv_math = conv_math
        N.check(N.lib().aev_net_set_conv_math(self.handle, {"f32": 0, "bf16x6": 1}[conv_math]), "aev_net_set_conv_math")
        self.names = ['intgr']
        for name, size in cnn_layers.items():
            if 'conv' in name:
                k = np.ascontiguousarray(weights['w_' + name], dtype=np.float32)
                b = np.ascontiguousarray(np.asarray(weights['b_' + name]).reshape(-1), dtype=np.float32)
                kh, kw, ci, co = k.shape
                N.check(N.lib().aev_net_add_conv(self.handle, k.ctypes.data_as(N.f32p), b.ctypes.data_as(N.f32p), kh, kw,
                                                 ci, co, 1, float(alpha), N.PAD[padding]), "aev_net_add_conv(%s)" % name)
            elif 'pool' in name:
                N.check(N.lib().aev_net_add_pool(self.handle, size[0], size[1], size[0]), "aev_net_add_pool(%s)" % name)
            else:
                raise NotImplementedError("non-event layer '%s' (the reference itself fails on these, SURVEY Q6)" % name)
            self.names.append(name)
        self.shapes = []
        for i in range(len(self.names)):
            chw = np.zeros(3, np.int32)
            N.check(N.lib().aev_layer_shape(self.handle, i, chw.ctypes.data_as(N.i32p)))
            self.shapes.append([int(v) for v in chw])
        self.kinds = ['intgr'] + ['conv' if 'conv' in nm else 'pool' for nm in self.names[1:]]
        self.conv_specs = [None] + [tuple(cnn_layers[nm]) for nm in self.names[1:]]

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            N.lib().aev_net_destroy(h)
            self.handle = None

    # ---- stepping -----------------------------------------------------------------------------
    def reset(self, stream=-1):
        N.check(N.lib().aev_net_reset(self.handle, int(stream)), "aev_net_reset")

    def step(self, batches):
        """batches: list of S int arrays [n_s,3] (y,x,ts) (an empty one leaves that stream untouched)."""
        if len(batches) != self.n_streams:
            raise ValueError("need one event batch per stream")
        off = np.zeros(self.n_streams + 1, np.int32)
        off[1:] = np.cumsum([len(b) for b in batches])
        rows = [np.asarray(b, dtype=np.int32).reshape(-1, 3) for b in batches]
        yxt = np.ascontiguousarray(np.concatenate(rows, 0)) if off[-1] else np.zeros((1, 3), np.int32)
        N.check(N.lib().aev_net_step(self.handle, yxt.ctypes.data_as(N.i32p), off.ctypes.data_as(N.i32p)), "aev_net_step")

    def queue_upload(self, steps):
        """steps: list (T) of lists (S) of [n,3] int arrays; uploaded once, then run without host round trips."""
        T = len(steps)
        lens = np.array([[len(b) for b in st] for st in steps], np.int64).reshape(-1)
        off = np.zeros(T * self.n_streams + 1, np.int64)
        off[1:] = np.cumsum(lens)
        flat = [np.asarray(b, dtype=np.int32).reshape(-1, 3) for st in steps for b in st]
        yxt = np.ascontiguousarray(np.concatenate(flat, 0)) if off[-1] else np.zeros((1, 3), np.int32)
        N.check(N.lib().aev_net_queue_upload(self.handle, yxt.ctypes.data_as(N.i32p), off.ctypes.data_as(N.i64p), T),
                "aev_net_queue_upload")
        self.queue_len, self.queue_events = T, int(off[-1])

    def queue_upload_samples(self, samples, example_hw, batch_event_size=200, batch_event_usec=None):
        """samples: list (S) of (x, y, ts) int arrays -- one WHOLE sample per stream, as the reader hands it to
        Runner.data_transform.  The device does what the reference's runner does on the host (runner.py:23-33,65-72): ts -= ts[0],
        center crop to the frame (utils.py:4-35) if `example_hw` differs from it, (y, x, ts) rows, chunks of ~batch_event_size
        events or time bins of batch_event_usec.  Returns the number of queue steps."""
        if len(samples) != self.n_streams:
            raise ValueError("need one sample per stream")
        lens = np.array([len(smp[0]) for smp in samples], np.int64)
        off = np.zeros(self.n_streams + 1, np.int64)
        off[1:] = np.cumsum(lens)
        cat = [np.ascontiguousarray(np.concatenate([np.asarray(smp[k]).reshape(-1) for smp in samples]).astype(np.int32)) for k in range(3)]
        T = C.c_int(0)
        N.check(N.lib().aev_net_queue_upload_samples(self.handle, cat[0].ctypes.data_as(N.i32p), cat[1].ctypes.data_as(N.i32p),
                                                     cat[2].ctypes.data_as(N.i32p), off.ctypes.data_as(N.i64p), int(example_hw[0]),
                                                     int(example_hw[1]), int(batch_event_size or 0), int(batch_event_usec or 0),
                                                     C.byref(T)), "aev_net_queue_upload_samples")
        self.queue_len = T.value
        # (the crop drops events: the number kept is the last offset of the queue the device has built)
        off = np.zeros(T.value * self.n_streams + 1, np.int64)
        N.check(N.lib().aev_net_queue_download(self.handle, None, 0, off.ctypes.data_as(N.i64p), C.byref(T)), "aev_net_queue_download")
        self.queue_events = int(off[-1])
        return self.queue_len

    def queue_download(self):
        """(steps, offsets int64 [T*S+1], yxt int32 [rows,3]) of the current event queue."""
        T = C.c_int(0)
        N.check(N.lib().aev_net_queue_download(self.handle, None, 0, None, C.byref(T)), "aev_net_queue_download")
        off = np.zeros(T.value * self.n_streams + 1, np.int64)
        N.check(N.lib().aev_net_queue_download(self.handle, None, 0, off.ctypes.data_as(N.i64p), C.byref(T)), "aev_net_queue_download")
        yxt = np.zeros((max(int(off[-1]), 1), 3), np.int32)
        N.check(N.lib().aev_net_queue_download(self.handle, yxt.ctypes.data_as(N.i32p), len(yxt), off.ctypes.data_as(N.i64p), C.byref(T)),
                "aev_net_queue_download")
        self.queue_events = int(off[-1])
        return T.value, off, yxt[:int(off[-1])]

    def queue_run(self, t0, T, use_graph=True):
        N.check(N.lib().aev_net_queue_run(self.handle, int(t0), int(T), 1 if use_graph else 0), "aev_net_queue_run")

    def queue_run_timed(self, t0, T, use_graph=True):
        ms = C.c_float(0)
        N.check(N.lib().aev_net_queue_run_timed(self.handle, int(t0), int(T), 1 if use_graph else 0, C.byref(ms)),
                "aev_net_queue_run_timed")
        return ms.value

    def queue_profile(self, t0, T):
        """Eager run with HIP events around every launch.  Returns (by_kernel, by_layer_kernel):
        {kernel: (total_ms, launches)} and {(layer_name, kernel): (total_ms, launches)}."""
        L = len(self.names)
        ms = np.zeros((L, N.NUM_KERNELS), np.float32)
        cnt = np.zeros((L, N.NUM_KERNELS), np.int64)
        N.check(N.lib().aev_net_queue_profile(self.handle, int(t0), int(T), ms.ctypes.data_as(N.f32p),
                                              cnt.ctypes.data_as(N.i64p)), "aev_net_queue_profile")
        names = [N.lib().aev_kernel_name(k).decode() for k in range(N.NUM_KERNELS)]
        by_kernel = {names[k]: (float(ms[:, k].sum()), int(cnt[:, k].sum())) for k in range(N.NUM_KERNELS) if cnt[:, k].sum()}
        by_layer = {(self.names[l], names[k]): (float(ms[l, k]), int(cnt[l, k]))
                    for l in range(L) for k in range(N.NUM_KERNELS) if cnt[l, k]}
        return by_kernel, by_layer

    def sync(self):
        N.check(N.lib().aev_net_sync(self.handle), "aev_net_sync")

    def set_dense(self, dense=True):
        """Dense-frame mode: every site of every layer is recomputed each step (BASELINE config 5 comparator).  True / 1:
        activation once (the event path's semantics); 2: activation again after every pool layer, as the reference's
        dense NumPy model does (frame_numpy.py:77-78); False / 0: back to the asynchronous mode."""
        N.check(N.lib().aev_net_set_dense(self.handle, int(dense)), "aev_net_set_dense")

    def set_conv_impl(self, impl):
        N.check(N.lib().aev_net_set_conv_impl(self.handle, {"auto": 0, "valu": 1, "mfma": 2}.get(impl, impl)))

    # ---- results ------------------------------------------------------------------------------
    def output(self):
        """float32 [S, H, W, C]: last layer's featuremap().transpose(1,2,0) per stream (event_numpy.py:79)."""
        c, h, w = self.shapes[-1]
        out = np.empty((self.n_streams, h, w, c), np.float32)
        N.check(N.lib().aev_net_output(self.handle, out.ctypes.data_as(N.f32p)), "aev_net_output")
        return out

    def output_into(self, device_ptr):
        """Write the same tensor into device memory (e.g. torch tensor .data_ptr()) -- the RCCL send buffer."""
        N.check(N.lib().aev_net_output_device(self.handle, C.c_void_p(int(device_ptr))), "aev_net_output_device")

    def detect(self, num_classes, num_bbox, conf_threshold=0.1, use_nms=True, nms_threshold=0.0):
        """YOLO decode + NMS of every stream's current output on the GPU (viz.py:125-164; the runner's thresholds,
        runner.py:36-39).  Returns float32 [S, nb, 7] = x, y, w, h, conf, class, valid."""
        c, h, w = self.shapes[-1]
        out = np.empty((self.n_streams, h * w * num_bbox, 7), np.float32)
        N.check(N.lib().aev_net_detect(self.handle, int(num_classes), int(num_bbox), float(conf_threshold), 1 if use_nms else 0,
                                       float(nms_threshold), out.ctypes.data_as(N.f32p)), "aev_net_detect")
        return out

    def detect_into(self, device_ptr, num_classes, num_bbox, conf_threshold=0.1, use_nms=True, nms_threshold=0.0):
        """Same, written to device memory [S, nb, 7] (the RCCL send buffer of the detection gather)."""
        N.check(N.lib().aev_net_detect_device(self.handle, int(num_classes), int(num_bbox), float(conf_threshold),
                                              1 if use_nms else 0, float(nms_threshold), C.c_void_p(int(device_ptr))),
                "aev_net_detect_device")

    def read(self, layer, which, stream=0):
        c, h, w = self.shapes[layer]
        if which == N.IDX_MAX:
            out = np.empty(c * h * w, np.int32)
        elif which == N.RECOMPUTE:
            out = np.empty((h, w), np.uint8)
        else:
            out = np.empty((c, h, w), np.float32)
        N.check(N.lib().aev_layer_read(self.handle, int(layer), int(stream), int(which), out.ctypes.data_as(C.c_void_p)),
                "aev_layer_read")
        return out

    def layer_events(self, layer, stream=0):
        n = C.c_int(0)
        N.check(N.lib().aev_layer_num_events(self.handle, layer, stream, C.byref(n)))
        y, x = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        N.check(N.lib().aev_layer_events(self.handle, layer, stream, y.ctypes.data_as(N.i32p), x.ctypes.data_as(N.i32p),
                                         n.value, C.byref(n)))
        return y, x

    def stats(self, reset=False):
        """(sites per layer, input events, steps) accumulated over all streams since the last reset."""
        sites = np.zeros(len(self.names), np.int64)
        ev, st = C.c_int64(0), C.c_int64(0)
        N.check(N.lib().aev_net_stats(self.handle, sites.ctypes.data_as(N.i64p), C.byref(ev), C.byref(st)))
        if reset:
            N.check(N.lib().aev_net_stats_reset(self.handle))
        return sites, ev.value, st.value

    def algorithmic_bytes_flops(self, sites, events_in, stream_steps):
        """bytes = D + sum_conv n(8 Cin k^2 + 8 Cout) + sum_pool n C (8 k^2 + 4) + 12 N_in, with
        D = (12 B x conv state elements + 8 B x H W) per stream-step; flops = sum_conv n * 2 * (2 Cout Cin k^2).
        `stream_steps` = number of (stream, step) pairs executed."""
        dense = 0
        by, fl = 0.0, 0.0
        for i, kind in enumerate(self.kinds):
            c, h, w = self.shapes[i]
            if kind == 'conv':
                kh, kw, ci, co = self.conv_specs[i]
                dense += 12 * c * h * w
                by += float(sites[i]) * (8 * ci * kh * kw + 8 * co)
                fl += float(sites[i]) * 2 * (2 * co * ci * kh * kw)
            elif kind == 'pool':
                kh, kw = self.conv_specs[i][:2]
                by += float(sites[i]) * c * (8 * kh * kw + 4)
        dense += 8 * self.h * self.w
        by += float(dense) * stream_steps + 12.0 * events_in
        return by, fl, float(dense)


class YoloEventHip:
    """Same constructor / build_graph surface as YoloEventNumpy (event_numpy.py:12-15,90-105); `weights` replaces
    (checkpoint, sess)."""

    def __init__(self, h_frame, w_frame, num_classes, cnn_layers, cnn_padding, h_cells, w_cells, num_bbox, alpha, leak,
                 weights, device=None):
        self._h_frame, self._w_frame, self._num_classes = h_frame, w_frame, num_classes
        self._cnn_layers = layers_dict(cnn_layers) if isinstance(cnn_layers, str) else cnn_layers
        self._padding, self._h_cells, self._w_cells, self._num_bbox = cnn_padding, h_cells, w_cells, num_bbox
        self._alpha, self._leak, self._weights, self._device = alpha, leak, weights, device

    def build_cnn_layers(self):
        from .layers.conv2d import Conv2DLayer
        from .layers.integration import IntegrationLayer
        from .layers.maxpool import MaxPoolLayer
        prev = IntegrationLayer(self._leak, self._h_frame, self._w_frame, self._device)
        event_layers, non_event_layers = [prev], []
        for name, size in self._cnn_layers.items():
            if 'conv' in name:
                prev = Conv2DLayer(prev, self._weights['w_' + name], self._weights['b_' + name], 1, self._alpha,
                                   self._padding)
                event_layers.append(prev)
            elif 'pool' in name:
                prev = MaxPoolLayer(prev, size, size[0])
                event_layers.append(prev)
            else:
                non_event_layers.append((name, size))
        return event_layers, non_event_layers

    def build_graph(self, _=None):
        event_layers, non_event_layers = self.build_cnn_layers()
        if non_event_layers:
            raise NotImplementedError("fc / flatten layers after the event layers (the reference fails on them, SURVEY Q6)")
        out_shape = [self._h_cells, self._w_cells, self._num_classes + self._num_bbox * 5]

        def graph(events, reset):
            if reset:
                for layer in event_layers:
                    layer.reset()
            event_layers[-1].compute_all(events, None)
            return np.reshape(event_layers[-1].featuremap().transpose(1, 2, 0), out_shape)

        graph.layers = event_layers
        return graph
