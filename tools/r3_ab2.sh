: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r3_ab2}; mkdir -p $O
A="--legs none --no-cpu-baseline --no-paint --steps 20 --warmup 5"
val() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms', d['value'], 'tiles/s')"; }
python bench.py $A 2>$O/a.err | val "f32 new"
BP_NOENC=1 BP_NOTHIN=1 python bench.py $A 2>$O/b.err | val "f32 old kernels"
python bench.py $A --dtype bf16 2>$O/c.err | val "bf16 new"
BP_NOENC=1 BP_NOTHIN=1 python bench.py $A --dtype bf16 2>$O/d.err | val "bf16 old kernels"
python tools/phase_times.py f32 2>/dev/null
python tools/phase_times.py bf16 2>/dev/null
