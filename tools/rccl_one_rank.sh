#!/bin/bash
# One rank drives the whole data-parallel schedule (BP_SYNC_FORCE=1): every batch-norm statistics all-reduce (float64)
# goes through the peer-memory kernel (csrc/peer_comm.hip; BP_PEER_SYNC=0: through RCCL like the gradients) and the
# flat gradient all-reduce is a real ncclAllReduce call with world size 1 -- default
# schedule (one communicator), the opt-in early all-reduce (BP_EARLY_ALLREDUCE=1: second communicator, weight-gradient
# stream) and throughput mode (--local-bn: no statistics collectives).  Not a scaling measurement: an API / ordering
# check of the N > 1 code path on a one-GPU box, and the per-collective latency floor with nothing to exchange.
export BP_SYNC_FORCE=1 MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
show() { python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']
print('$1', d['value'], 'tiles/s', d['ms_per_step'], 'ms/step', json.dumps({k: c.get(k) for k in ('batch_norm', 'collectives_per_step', 'gradient_bytes_per_step', 'inside_collectives', 'gradient_all_reduce', 'statistics_transport', 'backend')}))"; }
A="--legs none --steps 8 --warmup 3 --no-cpu-baseline --no-paint"
python bench.py $A 2>gpurun_out/rccl1.err | show "cvae f32 parity-mode"
BP_EARLY_ALLREDUCE=1 python bench.py $A 2>>gpurun_out/rccl1.err | show "cvae f32 early-allreduce"
python bench.py $A --local-bn 2>>gpurun_out/rccl1.err | show "cvae f32 local-bn"
python bench.py $A --dtype bf16 2>>gpurun_out/rccl1.err | show "cvae bf16 parity-mode"
python bench.py $A --dtype bf16 --local-bn 2>>gpurun_out/rccl1.err | show "cvae bf16 local-bn"
python bench.py --legs none --workload cgan --steps 2 --warmup 1 --no-cpu-baseline 2>>gpurun_out/rccl1.err | show "cgan"
tail -3 gpurun_out/rccl1.err
