#!/bin/bash
# one-time confidence checks of the round's kernel changes: smoke(), the fuzzers with more seeds (ONE run), the soak comparison
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/r3final; mkdir -p $O
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
AEV_FUZZ_SEEDS=120 python -m pytest tests/test_gpu_fuzz.py "tests/test_gpu_parity.py::test_random_architectures_many_streams_vs_oracle" "tests/test_gpu_parity.py::test_random_architectures_vs_oracle" -m gpu -q -p no:cacheprovider > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log
python tools/runs/soak.py > $O/soak.log 2>&1; echo "soak rc=$?"; tail -2 $O/soak.log
