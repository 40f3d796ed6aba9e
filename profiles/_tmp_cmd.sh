cd $GRAFT_REPO_ROOT
PYTHONPATH=$PWD timeout 900 python profiles/solve_soak.py 60 > gpurun_out/r04_solve_soak.txt 2>&1; cat gpurun_out/r04_solve_soak.txt
