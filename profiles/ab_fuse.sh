# A/B of the launch fusion switches on one box: kernel timelines under rocprofv3 and unprofiled
# bench lines (bash profiles/ab_fuse.sh; results under gpurun_out/)
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
FAST="--no-cpu-baseline --no-batched --no-whole-solve"
for v in "all:" "none:SLPX_FUSE_LAUNCHES=0"; do
  name=${v%%:*}; envs=${v#*:}
  rm -rf $O/prof_$name
  env $envs timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python bench.py $FAST --steps 100 --warmup 10 --repeats 2 > $O/ab_$name.log 2> $O/ab_$name.err
  KT=$(ls $O/prof_$name/*/*kernel_trace.csv | head -1)
  python profiles/timeline.py $KT > $O/timeline_$name.txt
  rm -rf $O/prof_$name
done
for rep in 1 2; do
  for v in "all:" "none:SLPX_FUSE_LAUNCHES=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs python bench.py $FAST 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])"
  done
done
