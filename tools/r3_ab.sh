# A/B of the deferred split-K reductions and the side-stream bf16 packs + single-layer timings of the few-channel layers
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3_ab}
mkdir -p $O
cd $R
A="--legs none --no-cpu-baseline --no-paint --steps 20 --warmup 5"
val() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms', d['value'], 'tiles/s')"; }
python bench.py $A 2>$O/a.err | val "f32 defer"
BP_DEFER_REDUCE=0 python bench.py $A 2>$O/b.err | val "f32 immediate"
python bench.py $A --dtype bf16 2>$O/c.err | val "bf16 defer"
BP_DEFER_REDUCE=0 python bench.py $A --dtype bf16 2>$O/d.err | val "bf16 immediate"
python tools/conv_bench.py 0,8,1,5,1,2,64,512,512 0,1,1,3,1,1,64,512,512 0,2,8,4,2,1,64,512,512 0,1,8,4,2,1,64,512,512 \
   0,8,16,8,4,2,64,256,256 0,16,32,8,4,2,64,64,64 1,1,1,8,4,2,64,128,128 2>&1 | tee $O/layers.txt
