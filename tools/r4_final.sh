# Round-4 evidence in ONE gpurun call, on the kernel sources as they are: everything lands under gpurun_out/r4_final/.
#   1. tools/final_round.sh (fp32 default + serial kernel stats, FETCH/WRITE PMC passes of BOTH dtypes -> pmc_traffic.json
#      keyed per dtype / kernel instance / launch shape with per-step sums, bf16 serial kernel stats, bench lines, per-layer tables)
#   2. paint-ONLY kernel stats (bench.py --workload paint: no training kernel in the process), CGAN kernel stats
#   3. matrix-core utilisation (SQ_VALU_MFMA_BUSY_CYCLES ...) of every kernel of the fp32 and the bf16 step
#   4. the default bench line (what the driver runs), phase timelines, the one-rank data-parallel drive (peer / RCCL)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/r4_final
rm -rf $F; mkdir -p $F
bash $R/tools/final_round.sh > $F/final_round.log 2>&1; echo "final_round rc=$?"
cp -r $R/gpurun_out/final/* $F/ 2>/dev/null
bash $R/tools/prof_paint.sh > $F/prof_paint.log 2>&1; echo "prof_paint rc=$?"
cp $R/gpurun_out/prof_paint/f32_summary.txt $F/paint_f32_kernel_stats_summary.txt
cp $R/gpurun_out/prof_paint/bf16_summary.txt $F/paint_bf16_kernel_stats_summary.txt
tail -n 1 $R/gpurun_out/prof_paint/f32.log > $F/bench_paint_f32.json; tail -n 1 $R/gpurun_out/prof_paint/bf16.log > $F/bench_paint_bf16.json
bash $R/tools/prof_cgan.sh > $F/prof_cgan.log 2>&1; echo "prof_cgan rc=$?"
cp $R/gpurun_out/prof_cgan/summary.txt $F/cgan_kernel_stats_summary.txt
cd /tmp && export TMPDIR=/tmp
export BP_SIDE_WGRAD=0 BP_BRANCH_STREAMS=0
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"
for d in f32 bf16; do
  timeout -k 10 400 rocprofv3 --pmc $PMC --kernel-trace -d $F/mfma_$d -o r --output-format csv -- python3 $R/bench.py --legs none --dtype $d --steps 2 --warmup 1 --no-cpu-baseline --no-paint > $F/mfma_$d.log 2>&1
  python3 $R/tools/mfma_util.py $(ls $F/mfma_$d/*counter_collection.csv | head -1) 30 > $F/mfma_util_$d.txt
  rm -rf $F/mfma_$d
  head -14 $F/mfma_util_$d.txt | cut -c1-150
done
unset BP_SIDE_WGRAD BP_BRANCH_STREAMS
cd $R
python bench.py --steps 20 --warmup 5 > $F/bench_default.json 2> $F/bench_default.err; echo "bench rc=$?"
tail -c 600 $F/bench_default.json
for d in f32 bf16; do python tools/phase_times.py $d 20 > $F/phase_times_$d.txt 2>/dev/null; cat $F/phase_times_$d.txt; done
bash tools/rccl_one_rank.sh > $F/dp_one_rank_peer.txt 2>&1
BP_PEER_SYNC=0 bash tools/rccl_one_rank.sh > $F/dp_one_rank_rccl.txt 2>&1
for l in k3 k4s2 t4s2; do LAYER=$l python tools/ws_bench.py 64 64 64 10 2>&1 | grep -v amdgpu.ids >> $F/ws_bench.txt; done
cut -c1-200 $F/dp_one_rank_peer.txt | head -6
