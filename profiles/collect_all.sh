# The whole evidence sequence of a round, on the GPU box, from the repo root:
#   bash profiles/collect_all.sh r03
# Every --pmc pass on its own, never combined with a trace (MI355X_MICROARCH.md).  Leaves the
# condensed files under gpurun_out/profiles_new/ (gpurun_out/ is what travels back).
set -x
TAG=${1:-r03}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
O=$R/gpurun_out
N=$O/profiles_new
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $N
mkdir -p $N
FAST="--no-cpu-baseline --no-batched --no-whole-solve"
# 0. unprofiled reference run of the exact default command
timeout 900 python bench.py > $O/bench_unprofiled.log 2> $O/bench_unprofiled.err
grep '^{' $O/bench_unprofiled.log | tail -1 > $N/${TAG}_bench_unprofiled.json
# 1. kernel stats of the default workload (single N=1000 problem; the after-the-region batch
#    probes and the whole solve are left out so that the averages are per-launch numbers of it)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py $FAST > $O/bench_profiled.log 2> $O/bench_profiled.err
grep '^{' $O/bench_profiled.log | tail -1 > $N/${TAG}_bench.json
# 2. counters, each in its own pass
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 20 --warmup 2 --repeats 1 --fixed-repeats $FAST > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 20 --warmup 2 --repeats 1 --fixed-repeats $FAST > $O/pmc_write.log 2>&1
python profiles/collect.py $TAG $O/prof $O/pmc_fetch $O/pmc_write > $O/collect.log 2>&1
KT=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python profiles/timeline.py $KT > $N/${TAG}_step_timeline.txt 2>> $O/collect.log
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_traffic.json $N/
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
# 3. the other configurations: kernel stats + counters each
for CFG in "N5000:--N 5000" "gfold:--workload gfold" "b64xN500:--workload batch512 --batch 64" "b512xN1000:--workload batch512 --batch 512 --N 1000"; do
  NAME=${CFG%%:*}
  ARGS=${CFG#*:}
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py $ARGS --steps 50 --warmup 5 --repeats 3 $FAST > $O/bench_$NAME.log 2> $O/bench_$NAME.err
  grep '^{' $O/bench_$NAME.log | tail -1 > $N/${TAG}_${NAME}_bench.json
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py $ARGS --steps 5 --warmup 1 --repeats 1 --fixed-repeats $FAST > $O/pmc_fetch_$NAME.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py $ARGS --steps 5 --warmup 1 --repeats 1 --fixed-repeats $FAST > $O/pmc_write_$NAME.log 2>&1
  python profiles/collect.py ${TAG}_$NAME $O/prof $O/pmc_fetch $O/pmc_write >> $O/collect.log 2>&1
  cp profiles/${TAG}_${NAME}_kernel_stats.csv profiles/${TAG}_${NAME}_traffic.json $N/
  rm -rf $O/prof $O/pmc_fetch $O/pmc_write
done
# 4. phase clocks inside the LDLT kernels and the latency microbenchmarks: COLLECT_CLOCKS=1 (r06 leaves the step kernel
# and the batch kernels as they were: the r05_* files of this section stand)
if [ "${COLLECT_CLOCKS:-0}" = "1" ]; then
set +x
bash profiles/ldlt_clocks.sh $TAG 1000 5000 > /dev/null 2>&1
cp $O/${TAG}_ldlt_clocks.txt $N/${TAG}_ldlt_clocks.txt
{
  D=$R/build/clocks_lib
  export PYTHONPATH=$R
  rm -f $R/tests/support/libslpx_models.so $R/tests/support/libslpx_hostcheck.so
  for c in 0 1; do SLPX_LIB=$D/libslpx.so LD_LIBRARY_PATH=$D python profiles/mf_timeline.py 1000 $c 2>&1 | grep -v "^slpx"; done
  SLPX_LIB=$D/libslpx.so LD_LIBRARY_PATH=$D python profiles/mf_timeline.py 5000 0 2>&1 | grep -v "^slpx"
  rm -f $R/tests/support/libslpx_models.so $R/tests/support/libslpx_hostcheck.so
} > $N/${TAG}_mf_timeline.txt 2>&1
{
  D=$R/build/clocks_lib
  rm -f $R/tests/support/libslpx_models.so $R/tests/support/libslpx_hostcheck.so
  for i in 1 2 3; do SLPX_LIB=$D/libslpx.so LD_LIBRARY_PATH=$D python profiles/chain_timeline.py 1000 300 2>&1 | grep -v "^slpx"; done
  rm -f $R/tests/support/libslpx_models.so $R/tests/support/libslpx_hostcheck.so
} > $N/${TAG}_chain_timeline.txt 2>&1
bash profiles/host_slack.sh 2 > $N/${TAG}_host_slack.txt 2>&1
PYTHONPATH=$R python profiles/mf_front_stats.py 1000 > $N/${TAG}_mf_front_stats.txt 2>&1
set -x
PYTHONPATH=$R python profiles/il_clocks.py 1000 512 > $N/${TAG}_il_clocks.txt 2>&1
for B in latency icache chain front; do [ -x profiles/microbench/${B}_bin ] && ./profiles/microbench/${B}_bin > $N/${TAG}_microbench_$B.txt 2>&1; done
fi
PYTHONPATH=$R python profiles/setup_time.py 1000 5000 100 300 500 2>&1 | grep -v "tape family\|row group\|chunk\|tape: " > $N/${TAG}_setup_time.txt
# 5. whole solves over the BASELINE horizons; the launch-fusion switches one by one
PYTHONPATH=$R timeout 600 python profiles/horizon_sweep.py > $N/${TAG}_horizon_sweep.txt 2>&1
bash profiles/ab_fuse.sh > $N/${TAG}_fusion_ab.txt 2>&1
for v in all none; do cp $O/timeline_$v.txt $N/${TAG}_step_timeline_fuse_$v.txt; done
# 6. the multifrontal step against the pair-list kernels it replaces, and the matrix-core path (g-fold)
{
  for W in "1000" "5000" "500" "gfold"; do
    echo "== $W: multifrontal (default)"; PYTHONPATH=$R python profiles/mf_time.py $W
    echo "== $W: pair lists (SLPX_LDLT_MF=0)"; SLPX_LDLT_MF=0 PYTHONPATH=$R python profiles/mf_time.py $W
  done
  echo "== gfold: update blocks on the matrix cores from 128 entries (SLPX_MFMA_MIN_ENTRIES=128)"
  SLPX_MFMA_MIN_ENTRIES=128 PYTHONPATH=$R python profiles/mf_time.py gfold
  echo "== gfold: never on the matrix cores (SLPX_MFMA_MIN_ENTRIES=1000000)"
  SLPX_MFMA_MIN_ENTRIES=1000000 PYTHONPATH=$R python profiles/mf_time.py gfold
  echo "== 1000: exact supernodes only (SLPX_RELAX_ZEROS=0)"; SLPX_RELAX_ZEROS=0 PYTHONPATH=$R python profiles/mf_time.py 1000
} > $N/${TAG}_mf_ab.txt 2>&1
# MFMA counters of the g-fold step with the matrix-core path on for every front of >= 128 entries: own pass, no trace
rm -rf $O/pmc_mfma
SLPX_MFMA_MIN_ENTRIES=128 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc_mfma -- python bench.py --workload gfold --steps 20 --warmup 2 --repeats 1 --fixed-repeats $FAST > $O/pmc_mfma.log 2>&1
python profiles/mfma_counters.py $O/pmc_mfma > $N/${TAG}_gfold_mfma.json 2>> $O/collect.log
rm -rf $O/pmc_mfma
# 7. the round's measured experiments that stayed behind switches, and the small-batch probe
bash profiles/chain_ab.sh > $N/${TAG}_chain_ab.txt 2>&1
if [ "${COLLECT_CLOCKS:-0}" = "1" ]; then
[ -x profiles/microbench/anyorder_bin ] && ./profiles/microbench/anyorder_bin > $N/${TAG}_microbench_anyorder.txt 2>&1
[ -x profiles/microbench/launch_gap_bin ] && ./profiles/microbench/launch_gap_bin > $N/${TAG}_microbench_launch_gap.txt 2>&1
fi
bash profiles/b64_probe.sh > $N/${TAG}_b64_probe.txt 2>&1
tail -5 $O/collect.log
# 8. (r04) the parity record of the timed kernels, the interior-point iteration: A/B of the look-ahead chain and
#    kernel stats + launch timeline of whole solves, the supernode-chain rules, the batch experiments, N=50
PYTHONPATH=$R timeout 900 python profiles/parity_errors.py > $N/${TAG}_parity_errors.txt 2>&1
bash profiles/solve_ab.sh > $N/${TAG}_solve_ab.txt 2>&1
bash profiles/solve_kernel_stats.sh 500 $O/solve_prof > /dev/null 2>&1
cp $O/solve_prof/solve500_kernel_stats.csv $N/${TAG}_solve500_kernel_stats.csv
cp $O/solve_prof/solve500_timeline.txt $N/${TAG}_solve500_timeline.txt
bash profiles/solve_kernel_stats.sh 1000 $O/solve_prof1000 > /dev/null 2>&1
cp $O/solve_prof1000/solve1000_kernel_stats.csv $N/${TAG}_solve1000_kernel_stats.csv
cp $O/solve_prof1000/solve1000_timeline.txt $N/${TAG}_solve1000_timeline.txt
# (r06) the common iteration with the host deciding (SLPX_IPM_PIPELINE=0) and with the device: launch by launch
bash profiles/pipeline_trace.sh 300 > /dev/null 2>&1
for v in 0 1; do cp $O/pipeline_trace_$v.txt $N/${TAG}_pipeline_trace_$v.txt; done
# (r06) the setup phase by phase (SLPX_SETUP_TIMING=1), warm caches
for NN in 1000 5000; do
  echo "== N=$NN"; SLPX_SETUP_TIMING=1 PYTHONPATH=$R python profiles/setup_time.py $NN 2>&1 | grep -v "tape family\|row group\|chunk\|tape: " | tail -60
done > $N/${TAG}_setup_phases.txt 2>&1
# (r05) the forward error of the step kernel over 24 seeded states; a new right-hand side through the fronts against the pair lists
PYTHONPATH=$R timeout 900 python profiles/forward_error_sweep.py gpu 1000 24 > $N/${TAG}_forward_error_sweep_gpu.txt 2>&1
PYTHONPATH=$R timeout 300 python profiles/mf_solve_time.py 100 500 1000 5000 > $N/${TAG}_mf_solve_time.txt 2>&1
PYTHONPATH=$R timeout 300 python profiles/product_sensitivity.py 50 > $N/${TAG}_product_sensitivity_N50.txt 2>&1
tail -3 $N/${TAG}_parity_errors.txt
