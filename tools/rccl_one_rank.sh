#!/bin/bash
# One rank drives the whole data-parallel schedule through RCCL (BP_SYNC_FORCE=1): every batch-norm statistics
# all-reduce (float64, main stream, first communicator) and the gradient all-reduces (weight-gradient stream, second
# communicator) are real ncclAllReduce calls with world size 1.  Not a scaling measurement: an API / ordering check
# of the N > 1 code path on a one-GPU box.
export BP_SYNC_FORCE=1 MASTER_ADDR=127.0.0.1
python bench.py --legs none --steps 4 --warmup 2 --no-cpu-baseline --no-paint 2>gpurun_out/rccl1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cvae f32', d['value'], d['ms_per_step'], json.dumps({k: d['config'].get(k) for k in ('collectives_per_step', 'gradient_bytes_per_step', 'ms_per_step_inside_collectives', 'backend')}))"
python bench.py --legs none --dtype bf16 --steps 4 --warmup 2 --no-cpu-baseline --no-paint 2>>gpurun_out/rccl1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cvae bf16', d['value'], d['ms_per_step'], json.dumps({k: d['config'].get(k) for k in ('collectives_per_step', 'ms_per_step_inside_collectives', 'backend')}))"
python bench.py --legs none --workload cgan --steps 2 --warmup 1 --no-cpu-baseline 2>>gpurun_out/rccl1.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cgan', d['value'], d['ms_per_step'])"
tail -3 gpurun_out/rccl1.err
