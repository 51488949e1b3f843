# SQ / LDS counters of fp32 igemm kernels on the thin strided layers (conv micro-benchmark), gpurun_out/pmc_f32/
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_f32
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/tools/conv_bench.py 0,16,32,4,2,1,64,512,512 1,32,16,4,2,1,64,256,256 0,16,8,7,1,3,64,512,512"
python3 $ARGS > $OUT/plain.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq -o r --output-format csv -- python3 $ARGS > $OUT/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/lds -o r --output-format csv -- python3 $ARGS > $OUT/lds.log 2>&1
cat $OUT/plain.log
python3 $R/tools/pmc_counters.py $(ls $OUT/sq/*counter_collection.csv | head -1) igemm
python3 $R/tools/pmc_counters.py $(ls $OUT/lds/*counter_collection.csv | head -1) igemm
