# Files gpurun_out/r4_final/ (tools/r4_final.sh, merged back by gpurun) under profiles/ with the round's prefix.
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/r4_final
for f in f32_default_kernel_stats_summary.txt f32_serial_kernel_stats_summary.txt bf16_serial_kernel_stats_summary.txt \
         f32_pmc_hbm_traffic_per_kernel.txt mfma_util_f32.txt mfma_util_bf16.txt paint_f32_kernel_stats_summary.txt \
         paint_bf16_kernel_stats_summary.txt cgan_kernel_stats_summary.txt phase_times_f32.txt phase_times_bf16.txt ws_bench.txt \
         f32_layers.txt bf16_layers.txt bench_default.json bench_f32.json bench_bf16.json bench_cgan.json bench_paint_f32.json \
         bench_paint_bf16.json f32_default_kernel_stats.csv f32_serial_kernel_stats.csv bf16_serial_kernel_stats.csv; do
  cp $F/$f profiles/r04_$f
done
cp $F/pmc_traffic.json profiles/pmc_traffic.json
cp $F/pmc_traffic.json profiles/r04_pmc_traffic.json
cp $F/dp_one_rank_peer.txt profiles/r04_rccl_one_rank.txt
cp $F/dp_one_rank_rccl.txt profiles/r04_rccl_one_rank_process_group.txt
python -c "
import json, sys
sys.path.insert(0, '.')
import bench
d = json.load(open('profiles/pmc_traffic.json'))
print('pmc stamp', d['source_hash'], 'sources', bench.source_hash(), 'MATCH' if d['source_hash'] == bench.source_hash() else 'STALE')"
