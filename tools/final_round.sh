# The round's evidence in one GPU call (run from the repo root through gpurun): rocprofv3 passes (profile_round.sh,
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
# prof_bf16.sh), then the benches the documents quote.  Everything lands under gpurun_out/final/.
set -e
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final
rm -rf $F; mkdir -p $F
bash $R/tools/profile_round.sh > $F/profile_round.log 2>&1
cp $R/gpurun_out/prof_round/pmc_traffic.json $F/pmc_traffic.json
cp $R/gpurun_out/prof_round/pmc_traffic.json $R/profiles/pmc_traffic.json      # (bench.py reads it for `traffic`)
cp $R/gpurun_out/prof_round/pmc_hbm_traffic_per_kernel.txt $F/f32_pmc_hbm_traffic_per_kernel.txt
cp $R/gpurun_out/prof_round/default_kernel_stats_summary.txt $F/f32_default_kernel_stats_summary.txt
cp $R/gpurun_out/prof_round/serial_kernel_stats_summary.txt $F/f32_serial_kernel_stats_summary.txt
cp $(ls $R/gpurun_out/prof_round/stats/*kernel_stats.csv | head -1) $F/f32_default_kernel_stats.csv
cp $(ls $R/gpurun_out/prof_round/stats_serial/*kernel_stats.csv | head -1) $F/f32_serial_kernel_stats.csv
echo "profile_round done"
bash $R/tools/prof_bf16.sh > $F/prof_bf16.log 2>&1
cp $R/gpurun_out/prof_bf16/summary.txt $F/bf16_serial_kernel_stats_summary.txt
cp $(ls $R/gpurun_out/prof_bf16/stats/*kernel_stats.csv | head -1) $F/bf16_serial_kernel_stats.csv
echo "prof_bf16 done"
cd $R
python bench.py --legs none > $F/bench_f32.json 2> $F/bench_f32.err
echo "bench f32 done"; cut -c1-160 $F/bench_f32.json
python bench.py --legs none --dtype bf16 > $F/bench_bf16.json 2> $F/bench_bf16.err
echo "bench bf16 done"; cut -c1-160 $F/bench_bf16.json
python bench.py --legs none --layers --no-cpu-baseline --no-paint > $F/f32_layers.txt 2>&1
python bench.py --legs none --layers --no-cpu-baseline --no-paint --dtype bf16 > $F/bf16_layers.txt 2>&1
echo "layers done"
python bench.py --legs none --workload cgan --steps 4 --warmup 1 --no-cpu-baseline > $F/bench_cgan.json 2> $F/bench_cgan.err
echo "cgan done"; cut -c1-160 $F/bench_cgan.json
