# SQ / LDS / TCC counters of the trunk weight-gradient kernels (conv micro-benchmark), gpurun_out/pmc_wgrad/
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_wgrad
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WHICH=w
ARGS="$R/tools/conv_bench_bf16.py 0 128 128 3 1 1 64 64 64 5"
python3 $ARGS > $OUT/plain.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq -o r --output-format csv -- python3 $ARGS > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/lds -o r --output-format csv -- python3 $ARGS > $OUT/lds.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o r --output-format csv -- python3 $ARGS > $OUT/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/tcc -o r --output-format csv -- python3 $ARGS > $OUT/tcc.log 2>&1
cat $OUT/plain.log
for d in sq lds fetch tcc; do python3 $R/tools/pmc_counters.py $(ls $OUT/$d/*counter_collection.csv | head -1) wgrad_ws; done
