cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_ipm_device_gpu.py tests/test_restoration_gpu.py -m gpu -x -q 2>&1 | tail -5)
for R in 1 2; do for V in "SLPX_IPM_LOOKAHEAD_RIDE=1" "SLPX_IPM_LOOKAHEAD_RIDE=0"; do echo "== $V"; for N in 100 500; do env $V PYTHONPATH=$PWD python profiles/solve_profile.py $N 2>&1 | grep "^$N" | awk '{print $1,$3,$4,$6}' | tr '\n' ';'; echo; done; done; done
