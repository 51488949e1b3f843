# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
#   kernel trace + stats of the default command (weight gradients overlapped on a second stream) and of the
#   serial schedule (BP_SIDE_WGRAD=0: every kernel alone on the GPU - the schedule bench.py times kernels in),
#   and the two HBM-traffic PMC passes (separate passes: TCC has 4 slots) on the serial schedule.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_round
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --legs none --steps 2 --warmup 1 --no-cpu-baseline --no-paint $BENCH_EXTRA"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o r --output-format csv -- python3 $ARGS > $OUT/stats.log 2>&1
export BP_SIDE_WGRAD=0 BP_BRANCH_STREAMS=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats_serial -o r --output-format csv -- python3 $ARGS > $OUT/stats_serial.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o r --output-format csv -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o r --output-format csv -- python3 $ARGS > $OUT/write.log 2>&1
# the same two passes for the bf16 step (its dominant kernel is priced against HBM)
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch_bf16 -o r --output-format csv -- python3 $ARGS --dtype bf16 > $OUT/fetch_bf16.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write_bf16 -o r --output-format csv -- python3 $ARGS --dtype bf16 > $OUT/write_bf16.log 2>&1
python3 $R/tools/pmc_to_json.py $OUT/pmc_traffic.json \
  f32:$(ls $OUT/fetch/*counter_collection.csv | head -1):$(ls $OUT/write/*counter_collection.csv | head -1) \
  bf16:$(ls $OUT/fetch_bf16/*counter_collection.csv | head -1):$(ls $OUT/write_bf16/*counter_collection.csv | head -1)
python3 $R/tools/pmc_summary.py $(ls $OUT/fetch/*counter_collection.csv | head -1) $(ls $OUT/write/*counter_collection.csv | head -1) > $OUT/pmc_hbm_traffic_per_kernel.txt
python3 $R/tools/prof_summary.py $(ls $OUT/stats/*kernel_stats.csv | head -1) 5 40 > $OUT/default_kernel_stats_summary.txt
python3 $R/tools/prof_summary.py $(ls $OUT/stats_serial/*kernel_stats.csv | head -1) 5 40 > $OUT/serial_kernel_stats_summary.txt
ls $OUT/*
