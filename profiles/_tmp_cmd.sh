cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do SLPX_TWIN_VERBOSE=1 PYTHONPATH=$PWD python profiles/solve_profile.py 100 > /tmp/o.txt 2>&1; echo "rc=$? $(grep -c '^100' /tmp/o.txt)"; done
bash profiles/gate_stamps.sh > gpurun_out/iteration_stamps.txt 2>&1
cat gpurun_out/iteration_stamps.txt
