cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_ipm_device_gpu.py tests/test_restoration_gpu.py tests/test_solve_pins.py -m gpu -x -q -s 2>&1 | grep -E "passed|failed|twin launches|launched ahead|Error" | tail -12)
for N in 50 100 300 500; do SLPX_TWIN_VERBOSE=1 PYTHONPATH=$PWD python profiles/solve_profile.py $N > /tmp/o.txt 2>&1; grep "twin attempts" /tmp/o.txt | tail -1 | cut -c1-420; grep "^$N" /tmp/o.txt | awk '{print $1,$3,$4,$6}' | tr '\n' ';'; echo; done
