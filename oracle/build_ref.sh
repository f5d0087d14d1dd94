#!/bin/sh
# Build the reference's own pysph/base/linalg3.pyx (no cyarray dependency) from
# where it lies under /root/reference into oracle/_ref/ -- used ONLY by
# tests/golden/make_golden.py for the eigen-decomposition inside
# MonaghanArtificialStress (pysph/sph/solid_mech/basic.py:162-242).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)/_ref; mkdir -p "$HERE"
REF=/root/reference/pysph/base
[ -f "$REF/linalg3.pyx" ] || { echo "reference not present"; exit 0; }
PYINC=$(python -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python -c "import numpy; print(numpy.get_include())")
EXT=$(python -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
cython -3 --cplus -I "$REF" "$REF/linalg3.pyx" -o "$HERE/linalg3.cpp"
g++ -O2 -fPIC -shared -ffp-contract=off -I"$PYINC" -I"$NPINC" -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION \
    "$HERE/linalg3.cpp" -o "$HERE/linalg3$EXT"
rm -f "$HERE/linalg3.cpp"
echo built "$HERE/linalg3$EXT"
