set -x
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
O=$R/gpurun_out
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/bprof $O/bpmc_fetch $O/bpmc_write
# unprofiled reference run
timeout 600 python bench.py > $O/bench_unprofiled.log 2>&1
# 1. kernel stats of the exact default command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py > $O/bench_profiled.log 2>&1
# 2. counters, each in its own pass
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-batched-roofline > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-batched-roofline > $O/pmc_write.log 2>&1
# 3. batch of 512
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bprof -- python bench.py --batched-roofline --no-cpu-baseline > $O/bbench_profiled.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bpmc_fetch -- python bench.py --batched-roofline --steps 2 --warmup 1 --no-cpu-baseline > $O/bpmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bpmc_write -- python bench.py --batched-roofline --steps 2 --warmup 1 --no-cpu-baseline > $O/bpmc_write.log 2>&1
# condense on the box (the raw traces are too big to travel)
python profiles/collect.py r01 $O/prof $O/pmc_fetch $O/pmc_write > $O/collect.log 2>&1
python profiles/collect.py r01_batched $O/bprof $O/bpmc_fetch $O/bpmc_write >> $O/collect.log 2>&1
KT=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python profiles/timeline.py $KT > $O/r01_step_timeline.txt 2>> $O/collect.log
mkdir -p $O/profiles_new && cp profiles/r01_kernel_stats.csv profiles/r01_traffic.json profiles/r01_batched_kernel_stats.csv profiles/r01_batched_traffic.json $O/profiles_new/ 
tail -1 $O/bench_profiled.log > $O/profiles_new/r01_bench.line
tail -1 $O/bench_unprofiled.log > $O/profiles_new/r01_bench_unprofiled.line
tail -1 $O/bbench_profiled.log > $O/profiles_new/r01_batched_bench.line
cp $O/r01_step_timeline.txt $O/profiles_new/
# keep the merged-back payload small
rm -rf $O/prof $O/pmc_fetch $O/pmc_write $O/bprof $O/bpmc_fetch $O/bpmc_write
cat $O/collect.log | tail -5
