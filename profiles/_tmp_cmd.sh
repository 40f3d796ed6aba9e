cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash profiles/collect_all.sh r04 > gpurun_out/collect_all.log 2>&1; tail -2 gpurun_out/collect_all.log | cut -c1-200
bash profiles/gate_stamps.sh > gpurun_out/iteration_stamps.txt 2>&1; head -4 gpurun_out/iteration_stamps.txt | cut -c1-200
