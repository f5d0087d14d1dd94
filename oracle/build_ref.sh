#!/bin/sh
# Build the two Cython modules of the reference on the path that have no cyarray dependency from where they lie
# under /root/reference into oracle/_ref/ (outputs only; never copied sources):
#   pysph/base/linalg3.pyx    used by tests/golden/make_golden.py for the eigen-decomposition inside
#                             MonaghanArtificialStress (pysph/sph/solid_mech/basic.py:162-242)
#   pysph/base/c_kernels.pyx  the reference's COMPILED smoothing kernels (what its Cython backend evaluates): a second
#                             pin of the oracle's kernels next to the golden vectors made from the Python classes
#                             (tests/test_oracle_golden.py::test_oracle_kernels_vs_compiled_reference)
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
cython -3 --cplus -I "$REF" "$REF/c_kernels.pyx" -o "$HERE/c_kernels.cpp"
g++ -O2 -fPIC -shared -ffp-contract=off -I"$PYINC" -I"$NPINC" -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION \
    "$HERE/c_kernels.cpp" -o "$HERE/c_kernels$EXT"
rm -f "$HERE/c_kernels.cpp"
echo built "$HERE/c_kernels$EXT"
