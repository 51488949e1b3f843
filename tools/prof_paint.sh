# rocprofv3 kernel trace of the paint_stream leg ALONE (bench.py --workload paint: no training step runs in the process; fp32
# and bf16 trunk), summaries per 128-tile batch to gpurun_out/prof_paint/
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_paint
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in f32 bf16; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/$d -o r --output-format csv -- python3 $R/bench.py --workload paint --dtype $d --paint-batch 128 --paint-tiles 1024 > $OUT/$d.log 2>&1
  # (per 128-tile batch: 2 capture batches + 8 streamed + 10 resident replays = 20 batches)
  python3 $R/tools/prof_summary.py $(ls $OUT/$d/*kernel_stats.csv | head -1) 20 30 > $OUT/${d}_summary.txt
done
cat $OUT/f32_summary.txt | head -12
