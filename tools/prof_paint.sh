# rocprofv3 kernel trace of the paint_stream leg alone (fp32 and bf16 trunk), summaries to gpurun_out/prof_paint/
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_paint
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in f32 bf16; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/$d -o r --output-format csv -- python3 $R/bench.py --legs none --dtype $d --steps 1 --warmup 1 --no-cpu-baseline --paint-tiles 512 > $OUT/$d.log 2>&1
  python3 $R/tools/prof_summary.py $(ls $OUT/$d/*kernel_stats.csv | head -1) 2 25 > $OUT/${d}_summary.txt
done
cat $OUT/f32_summary.txt | head -12
