# The interior-point iteration launch by launch with the host deciding every iteration (SLPX_IPM_PIPELINE=0) and with the
# common iteration decided on the device, the next step enqueued ahead (default): bash profiles/pipeline_trace.sh 300
NN=${1:-300}
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
for v in 0 1; do
  rm -rf /tmp/pipe_tr$v
  SLPX_IPM_PIPELINE=$v PYTHONPATH=$PWD rocprofv3 --kernel-trace --output-format csv -d /tmp/pipe_tr$v -- python profiles/solve_profile.py $NN > $O/pipeline_trace_log$v.txt 2>&1
  f=$(ls /tmp/pipe_tr$v/*/*kernel_trace.csv | head -1)
  python profiles/pipeline_trace.py $f > $O/pipeline_trace_$v.txt
  rm -rf /tmp/pipe_tr$v
done
