#!/usr/bin/env bash
# TEST INFRASTRUCTURE — not part of the product path.
#
# Builds the reference's one native module (src/libs/cutils.pyx, Cython) from
# the sources where they lie under /root/reference into oracle/_ref/ so that the
# reference's own Python layers (src/layers/*.py) can be imported IN THIS
# CONTAINER to (a) validate the C restatement in oracle/aev_oracle.c and
# (b) generate the golden vectors under tests/golden/ (tests/golden/gen_golden.py).
#
# Recipe = what the reference's cython_setup.py:8-18 does (one Extension,
# -fopenmp, no NPY_NO_DEPRECATED_API), driven by cython + gcc directly.
# Nothing is copied into the repo: the generated .c is deleted after the
# compile and oracle/_ref/ is git-ignored.  The GPU box has no /root/reference;
# there this script is a no-op and nothing at run time depends on its output.
set -euo pipefail
REF=${AEV_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -f "$REF/src/libs/cutils.pyx" ]; then
  echo "build_ref: $REF not present - skipping (expected on the GPU box)"; exit 0
fi
mkdir -p "$OUT/src/libs"
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
TMPC=$(mktemp -d)
trap 'rm -rf "$TMPC"' EXIT
cython -3 "$REF/src/libs/cutils.pyx" -o "$TMPC/cutils.c" 2>/dev/null
gcc -O2 -fPIC -shared -fopenmp -w -I"$PYINC" -I"$NPINC" "$TMPC/cutils.c" -o "$OUT/src/libs/cutils$EXT"
echo "build_ref: built $OUT/src/libs/cutils$EXT"
