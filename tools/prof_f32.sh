# rocprofv3 kernel trace of the bf16 step (serial schedule), summary to gpurun_out/prof_f32/
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_f32
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BP_SIDE_WGRAD=0 BP_BRANCH_STREAMS=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python3 $R/bench.py --legs none --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-paint > $OUT/stats.log 2>&1
python3 $R/tools/prof_summary.py $(ls $OUT/stats/*kernel_stats.csv | head -1) 6 60 > $OUT/summary.txt
cat $OUT/summary.txt
