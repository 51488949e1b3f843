# rocprofv3 kernel trace of the CGAN iteration, summary to gpurun_out/prof_cgan/
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_cgan
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python3 $R/bench.py --legs none --workload cgan --steps 2 --warmup 1 --no-cpu-baseline --no-paint > $OUT/stats.log 2>&1
python3 $R/tools/prof_summary.py $(ls $OUT/stats/*kernel_stats.csv | head -1) 3 40 > $OUT/summary.txt
cat $OUT/summary.txt
