#!/bin/bash
O=gpurun_out/r2_final; mkdir -p $O
for i in 1 2; do python -m pytest tests -m gpu -x -q > $O/full_$i.log 2>&1; echo "full $i rc=$? $(tail -1 $O/full_$i.log | cut -c1-70)"; done
for i in 1 2 3 4; do AEV_FUZZ_SEEDS=120 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "random_arch" > $O/fz_$i.log 2>&1; echo "fuzz run $i rc=$? $(tail -1 $O/fz_$i.log | cut -c1-70)"; grep -h "AssertionError:" $O/fz_$i.log | cut -c1-600; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/runs/r2_profiles.sh r02b > $O/profiles.log 2>&1; tail -4 $O/profiles.log
