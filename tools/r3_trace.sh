# Kernel traces of the DEFAULT (overlapped) schedule, both dtypes, with the timeline report of tools/overlap_report.py
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-r3_trace}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in f32 bf16; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/$d -o r --output-format csv -- python3 $R/bench.py --legs none --dtype $d --steps 3 --warmup 1 --no-cpu-baseline --no-paint > $OUT/$d.log 2>&1
  python3 $R/tools/overlap_report.py $OUT/$d/r_kernel_trace.csv > $OUT/${d}_overlap.txt
  python3 $R/tools/prof_summary.py $OUT/$d/r_kernel_stats.csv 6 70 > $OUT/${d}_default_summary.txt
  cat $OUT/${d}_overlap.txt
done
