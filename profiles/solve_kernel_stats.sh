# rocprofv3 kernel stats + timeline tail of three whole solves at N (default 500): bash profiles/solve_kernel_stats.sh 500 out_dir
N=${1:-500}; O=${2:-gpurun_out/solve_prof}
export TMPDIR=/tmp
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python profiles/solve_profile.py $N > $O/solve_prof.log 2>&1
cp $O/prof/*/*kernel_stats.csv $O/solve${N}_kernel_stats.csv
python profiles/solve_timeline.py $(ls $O/prof/*/*kernel_trace.csv | head -1) > $O/solve${N}_timeline.txt
rm -rf $O/prof
