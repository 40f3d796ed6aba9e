export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
FAST="--no-cpu-baseline --no-batched --no-whole-solve"
for f in 1 0; do
  rm -rf $O/prof$f
  SLPX_FUSE_LAUNCHES=$f timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$f -- python bench.py $FAST --steps 100 --warmup 10 --repeats 2 > $O/ab$f.log 2> $O/ab$f.err
  KT=$(ls $O/prof$f/*/*kernel_trace.csv | head -1)
  python profiles/timeline.py $KT > $O/timeline_fuse$f.txt
  ST=$(ls $O/prof$f/*/*kernel_stats.csv | head -1)
  head -8 $ST > $O/stats_fuse$f.csv
  rm -rf $O/prof$f
done
for f in 1 0 1 0; do SLPX_FUSE_LAUNCHES=$f python bench.py $FAST 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse=$f', d['value'], d['ms_per_step'])"; done
