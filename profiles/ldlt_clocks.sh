#!/bin/bash
# Phase clocks of the step kernel of the fronts, from a library built with -DSLPX_MF_CLOCKS (the clocks stay in LDS
# until the task is through: ldlt_mf_kernels.h) -> gpurun_out/<tag>_ldlt_clocks.txt
#   bash profiles/ldlt_clocks.sh [tag] [N ...]
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r05}
shift || true
NS=${*:-1000}
export PYTHONPATH=$R
# (the test-support libraries link the libslpx they were built beside: they are rebuilt for the library under test
# and removed afterwards, so that the next user builds them against the tree's again)
trap 'rm -f $R/tests/support/libslpx_models.so $R/tests/support/libslpx_hostcheck.so' EXIT
rm -f $R/tests/support/libslpx_models.so $R/tests/support/libslpx_hostcheck.so
mkdir -p $R/gpurun_out
# (the models' library names libslpx.so as a dependency: the instrumented build carries the same file name in a
# directory of its own, found first through LD_LIBRARY_PATH, so that the process holds ONE libslpx)
D=$R/build/clocks_lib
if [ ! -f $D/libslpx.so ] || [ $R/sleipnir_amd/csrc/ldlt_mf_kernels.h -nt $D/libslpx.so ]; then
  mkdir -p $D
  make -C $R/sleipnir_amd/csrc -j8 OUT=$D/libslpx.so BUILD=../../build/slpx_clocks \
    CXXFLAGS="-O3 -std=c++23 -fPIC -Wall -Wno-unused-function -Wno-unused-result -DSLPX_MF_CLOCKS" > /dev/null || exit 1
  cp -r $R/sleipnir_amd/jit_cache $D/ 2>/dev/null
fi
export SLPX_LIB=$D/libslpx.so LD_LIBRARY_PATH=$D:${LD_LIBRARY_PATH:-}
O=$R/gpurun_out/${TAG}_ldlt_clocks.txt
: > $O
for N in $NS; do
  for w in 0 1 18 60 100; do
    python $R/profiles/ldlt_clocks.py $N $w 2>&1 | grep -v "^slpx\|level loop" >> $O
  done
done
cat $O
