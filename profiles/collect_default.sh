set -x
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; N=$O/profiles_new; TAG=r03
rm -rf $O/prof $O/pmc_fetch $O/pmc_write; mkdir -p $N
FAST="--no-cpu-baseline --no-batched --no-whole-solve"
# the default workload's evidence alone (collect_all.sh steps 0-2): unprofiled default command, kernel trace, counters
timeout 900 python bench.py > $O/bench_unprofiled.log 2> $O/bench_unprofiled.err
grep '^{' $O/bench_unprofiled.log | tail -1 > $N/${TAG}_bench_unprofiled.json
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py $FAST > $O/bench_profiled.log 2> $O/bench_profiled.err
grep '^{' $O/bench_profiled.log | tail -1 > $N/${TAG}_bench.json
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 20 --warmup 2 --repeats 1 --fixed-repeats $FAST > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 20 --warmup 2 --repeats 1 --fixed-repeats $FAST > $O/pmc_write.log 2>&1
grep '^{' $O/pmc_fetch.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('under --pmc', d['value'], d['ms_per_step'])"
python profiles/collect.py $TAG $O/prof $O/pmc_fetch $O/pmc_write
KT=$(ls $O/prof/*/*kernel_trace.csv | head -1)
python profiles/timeline.py $KT > $N/${TAG}_step_timeline.txt
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_traffic.json $N/
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
