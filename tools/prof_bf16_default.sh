# rocprofv3 kernel trace of the bf16 step in its DEFAULT (multi-stream) schedule + the overlap report, to gpurun_out/prof_bf16_default/
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bf16_default
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python3 $R/bench.py --legs none --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-paint > $OUT/stats.log 2>&1
python3 $R/tools/prof_summary.py $(ls $OUT/stats/*kernel_stats.csv | head -1) 7 40 > $OUT/summary.txt
python3 $R/tools/overlap_report.py $(ls $OUT/stats/*kernel_trace.csv | head -1) > $OUT/overlap.txt
cat $OUT/overlap.txt
