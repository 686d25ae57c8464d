import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if os.environ.get("AEV_TEST_TORCH_FIRST"):  # diagnostic only (DESIGN.md "HIP runtime"): run the suite on torch's bundled HIP runtime
    import torch  # noqa: F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    """Probe through the library itself (aev_device_count), never through torch: importing torch first would map its
    bundled HIP runtime (ROCm 7.0) under a library built by /opt/rocm's hipcc (7.2), see DESIGN.md "HIP runtime"."""
    try:
        from async_ev_cnn_amd import _native
        return _native.device_count() > 0
    except Exception:
        return False


def pytest_report_header(config):
    try:
        from async_ev_cnn_amd import _native
        i = _native.runtime_info()
        return "aev: HIP_VERSION built %d, runtime %d (%s), driver %d" % (i["hip_compiled"], i["hip_runtime"], i["runtime_path"], i["hip_driver"])
    except Exception as e:  # (library not built yet)
        return "aev: library not loadable: %s" % e


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
