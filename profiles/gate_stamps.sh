# Wall clocks between the kernels of an interior-point iteration, from inside the kernels (-DSLPX_GATE_STAMPS:
# ldlt_kernels.h, DeviceNlp::debug_gate_stamps), default and with the step launched ahead (SLPX_PRELAUNCH=1).
# Puts the stamped build (build/libslpx_stamps.so) in the in-tree library's place while it runs.
#   bash profiles/gate_stamps.sh > gpurun_out/iteration_stamps.txt
[ -f build/libslpx_stamps.so ] || make -C sleipnir_amd/csrc -j8 BUILD=../../build/slpx_stamps OUT=../../build/libslpx_stamps.so \
  CXXFLAGS="-O3 -std=c++23 -fPIC -Wall -Wno-unused-function -Wno-unused-result -DSLPX_GATE_STAMPS" > /dev/null 2>&1
cp sleipnir_amd/libslpx.so /tmp/libslpx_plain.so
cp build/libslpx_stamps.so sleipnir_amd/libslpx.so
trap 'cp /tmp/libslpx_plain.so sleipnir_amd/libslpx.so' EXIT
for V in "SLPX_PRELAUNCH=0" "SLPX_PRELAUNCH=1" "SLPX_PRELAUNCH=0" "SLPX_PRELAUNCH=1"; do
  echo "== $V"
  for N in 100 500; do
    env $V SLPX_TWIN_VERBOSE=1 PYTHONPATH=$PWD python profiles/solve_profile.py $N > /tmp/stamps.txt 2>&1
    grep "gate stamps" /tmp/stamps.txt | tail -2 | sed 's/gate -> counters out [0-9.]*, //; s/; the step before.*//'
    grep "^$N" /tmp/stamps.txt | awk '{print "   N", $1, "iterations", $3, "factorizations", $4, "t_total", $6}' | tail -1
  done
done
