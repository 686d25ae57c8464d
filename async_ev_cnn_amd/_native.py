"""ctypes binding of libaev_hip.so (include/aev.h).  Fails loudly when the library is missing."""
import ctypes as C
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_DIR, "libaev_hip.so")
CSRC = os.path.join(_DIR, "csrc")

SURFACE, LAYER_ACTFN, CONV_ACTFN, FEATUREMAP, RAW_ACT, IDX_MAX, RECOMPUTE = range(7)
PAD = {"VALID": 0, "SAME": 1}
NUM_KERNELS = 12

_ERRORS = {-1: NotImplementedError, -2: ValueError, -3: IndexError, -4: MemoryError, -5: ValueError,
           -6: RuntimeError, -7: NotImplementedError}

i32p, i64p, f32p, vp = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_void_p

# name -> (restype, argtypes): every symbol include/aev.h declares
PROTOTYPES = {
    "aev_last_error": (C.c_char_p, []),
    "aev_version": (C.c_int, []),
    "aev_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "aev_runtime_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "aev_device_sync": (C.c_int, [C.c_int]),
    "aev_im2col_event": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, i32p, i32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, f32p, i32p, i32p, C.POINTER(C.c_int), C.c_int]),
    "aev_min_argmax": (C.c_int, [f32p, f32p, C.c_int, C.c_int, i32p, i32p, C.c_int]),
    "aev_conv2d_event": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, i32p, i32p, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   f32p, i32p, i32p, C.POINTER(C.c_int), C.c_int]),
    "aev_conv2d": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_int, C.c_int, f32p, C.c_int]),
    "aev_im2col": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]),
    "aev_net_create": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(vp)]),
    "aev_net_destroy": (None, [vp]),
    "aev_net_add_conv": (C.c_int, [vp, f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
    "aev_net_add_pool": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    "aev_net_num_layers": (C.c_int, [vp]),
    "aev_net_num_streams": (C.c_int, [vp]),
    "aev_layer_shape": (C.c_int, [vp, C.c_int, i32p]),
    "aev_net_reset": (C.c_int, [vp, C.c_int]),
    "aev_layer_reset": (C.c_int, [vp, C.c_int, C.c_int]),
    "aev_layer_step": (C.c_int, [vp, C.c_int, C.c_int, i32p, i32p, C.c_int, C.POINTER(C.c_double)]),
    "aev_layer_num_events": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "aev_layer_events": (C.c_int, [vp, C.c_int, C.c_int, i32p, i32p, C.c_int, C.POINTER(C.c_int)]),
    "aev_layer_read": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    "aev_net_step": (C.c_int, [vp, i32p, i32p]),
    "aev_net_queue_upload": (C.c_int, [vp, i32p, i64p, C.c_int]),
    "aev_net_queue_upload_samples": (C.c_int, [vp, i32p, i32p, i32p, i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "aev_net_queue_download": (C.c_int, [vp, i32p, C.c_longlong, i64p, C.POINTER(C.c_int)]),
    "aev_net_queue_run": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    "aev_net_queue_run_timed": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "aev_net_queue_profile": (C.c_int, [vp, C.c_int, C.c_int, f32p, i64p]),
    "aev_kernel_name": (C.c_char_p, [C.c_int]),
    "aev_net_sync": (C.c_int, [vp]),
    "aev_net_output": (C.c_int, [vp, f32p]),
    "aev_net_output_device": (C.c_int, [vp, vp]),
    "aev_net_stats": (C.c_int, [vp, i64p, i64p, i64p]),
    "aev_net_stats_reset": (C.c_int, [vp]),
    "aev_net_set_conv_impl": (C.c_int, [vp, C.c_int]),
    "aev_net_set_dense": (C.c_int, [vp, C.c_int]),
    "aev_net_set_tuning": (C.c_int, [vp, C.c_int]),
    "aev_net_set_conv_math": (C.c_int, [vp, C.c_int]),
    "aev_yolo_detect": (C.c_int, [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                                  f32p, i32p, i32p, C.c_int]),
    "aev_nms": (C.c_int, [f32p, f32p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_float, i32p, i32p, C.c_int]),
    "aev_decode_nformat": (C.c_int, [C.POINTER(C.c_uint8), C.c_longlong, i32p, i32p, i32p, i32p, C.POINTER(C.c_longlong), C.c_int]),
    "aev_net_detect": (C.c_int, [vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, f32p]),
    "aev_net_detect_device": (C.c_int, [vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, vp]),
    "aev_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "aev_comm_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.POINTER(vp)]),
    "aev_comm_destroy": (None, [vp]),
    "aev_comm_rank": (C.c_int, [vp]),
    "aev_comm_world": (C.c_int, [vp]),
    "aev_comm_allreduce": (C.c_int, [vp, C.POINTER(C.c_double), C.c_int, C.c_int]),
    "aev_net_gather_detections": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_int, f32p, i32p, C.POINTER(C.c_int)]),
}

_LIB = None


def build(force=False):
    """Compile libaev_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(os.path.dirname(_DIR), "include", "aev.h"))
    stale = force or not os.path.exists(SO_PATH) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-B" if force else "-s"])
    return SO_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "libaev_hip.so is missing (%s). Build it with `make -C %s` or __graft_entry__.build(); "
                "there is no CPU fallback for this path." % (SO_PATH, CSRC))
        L = C.CDLL(SO_PATH)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(L, name)  # AttributeError here = ABI mismatch: fail loudly
            f.restype, f.argtypes = res, args
        _LIB = L
    return _LIB


def check(rc, what=""):
    if rc < 0:
        msg = lib().aev_last_error().decode("utf-8", "replace")
        raise _ERRORS.get(rc, RuntimeError)("%s%s" % (what + ": " if what else "", msg))
    return rc


def device_count():
    """Usable HIP devices, 0 when there is none (no exception: this is the GPU probe of the tests and of bench.py --
    it goes through the library, not through torch, so that the HIP runtime the library was built for is the one mapped)."""
    n = C.c_int(0)
    lib().aev_device_count(C.byref(n))
    return n.value


def device_sync(device=0):
    check(lib().aev_device_sync(int(device)), "aev_device_sync")


def runtime_info():
    """{'hip_compiled', 'hip_runtime', 'hip_driver', 'runtime_path', 'matched'}: HIP_VERSION of the build against the
    runtime this process has mapped (torch's bundled one if torch was imported first, /opt/rocm's otherwise)."""
    a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
    buf = C.create_string_buffer(1024)
    lib().aev_runtime_info(C.byref(a), C.byref(b), C.byref(c), buf, 1024)
    return {"hip_compiled": a.value, "hip_runtime": b.value, "hip_driver": c.value,
            "runtime_path": buf.value.decode("utf-8", "replace"),
            "matched": a.value // 100000 == b.value // 100000}


def default_device():
    """LOCAL_RANK-aware default device (one process per GPU under torch.distributed.run)."""
    if "AEV_DEVICE" in os.environ:
        return int(os.environ["AEV_DEVICE"])
    n = C.c_int(0)
    check(lib().aev_device_count(C.byref(n)), "aev_device_count")
    return int(os.environ.get("LOCAL_RANK", "0")) % max(n.value, 1)
