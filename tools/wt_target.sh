#!/bin/bash
# workgroup-count target of wgrad_tiles(_dma)_kernel (split-K partitions): whole step (overlapped) and kernels alone
for t in 320 384 448 512 640 768; do
  BP_WT_TARGET=$t python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-paint 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('target $t step', d['value'], d['ms_per_step'])"
done
for t in 384 512; do echo "target $t alone"; BP_WT_TARGET=$t bash tools/wt_bench.sh; done
