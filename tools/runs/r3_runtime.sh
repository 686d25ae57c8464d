#!/bin/bash
# Round 3, first GPU call: which HIP runtime serves the library, and do the two runtime work-arounds (stream pool, kernel
# warm-up) matter on the matched one?  Every suite runs ONCE.
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/r3_rt; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || exit 1
python - > $O/runtime.txt 2>&1 <<'PY'
from async_ev_cnn_amd import _native as N
print("library first :", N.runtime_info())
PY
python - >> $O/runtime.txt 2>&1 <<'PY'
import torch
from async_ev_cnn_amd import _native as N
print("torch first   :", N.runtime_info(), "torch.version.hip", torch.version.hip)
PY
cat $O/runtime.txt
# C: the canonical suite (matched runtime, work-arounds on)
python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/suite_default.log 2>&1; echo "default suite rc=$?" | tee -a $O/summary.txt
tail -25 $O/suite_default.log
# A: matched runtime, work-arounds OFF
AEV_NO_STREAM_POOL=1 AEV_NO_WARM=1 AEV_SEGV_TRACE=$R/$O/segv_matched_off.txt python -m pytest tests -m gpu -q -p no:cacheprovider > $O/suite_matched_off.log 2>&1; echo "matched runtime, work-arounds off rc=$?" | tee -a $O/summary.txt
tail -5 $O/suite_matched_off.log

# B: torch's bundled runtime (7.0) under the 7.2-built library, work-arounds OFF -- last: a host-side crash is the expected outcome
AEV_TEST_TORCH_FIRST=1 AEV_NO_STREAM_POOL=1 AEV_NO_WARM=1 AEV_SEGV_TRACE=$R/$O/segv_torch_off.txt python -m pytest tests -m gpu -q -p no:cacheprovider > $O/suite_torch_off.log 2>&1; echo "torch runtime, work-arounds off rc=$?" | tee -a $O/summary.txt
tail -5 $O/suite_torch_off.log
cat $O/summary.txt
