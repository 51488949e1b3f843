#!/bin/bash
python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-paint --dtype bf16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'])
pk = d['roofline']['per_kernel']
for k in ('bn_backward_apply_bf16_kernel', 'act_backward_bf16_kernel', 'act_backward_fast_kernel', 'bn_backward_apply_fast_kernel'):
    print(k, pk.get(k))"
