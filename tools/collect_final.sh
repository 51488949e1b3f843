#!/bin/bash
# after `gpurun -- bash tools/r3_final.sh`: copy the merged evidence from gpurun_out/final/ into profiles/ (tracked)
F=${2:-gpurun_out/r3_final}; R=${1:-r03}
cp $F/pmc_traffic.json profiles/pmc_traffic.json
for n in bench_f32.json bench_bf16.json bench_cgan.json f32_layers.txt bf16_layers.txt f32_pmc_hbm_traffic_per_kernel.txt \
         f32_default_kernel_stats_summary.txt f32_serial_kernel_stats_summary.txt f32_default_kernel_stats.csv \
         f32_serial_kernel_stats.csv bf16_serial_kernel_stats_summary.txt bf16_serial_kernel_stats.csv \
         bench_default.json paint_f32_kernel_stats_summary.txt paint_bf16_kernel_stats_summary.txt \
         cgan_kernel_stats_summary.txt mfma_util_f32.txt mfma_util_bf16.txt; do
  cp $F/$n profiles/${R}_$n
done
python -c "
import bench, json; print('sources', bench.source_hash(), 'stamp', json.load(open('profiles/pmc_traffic.json'))['source_hash'])"
