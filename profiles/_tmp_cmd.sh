cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/suite.txt 2>&1
(SLPX_FUSE_LAUNCHES=0 timeout 900 python -m pytest tests/test_slp_surface.py tests/test_restoration_gpu.py -m gpu -x -q 2>&1 | tail -40) > gpurun_out/fuse0.txt 2>&1
for N in 50 100 300 500; do SLPX_TWIN_VERBOSE=1 PYTHONPATH=$PWD python profiles/solve_profile.py $N 2>&1 | tail -2; done > gpurun_out/twin_hist.txt 2>&1
cat gpurun_out/suite.txt gpurun_out/fuse0.txt gpurun_out/twin_hist.txt
