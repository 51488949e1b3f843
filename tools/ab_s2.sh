#!/bin/bash
# A/B of the stride-2 flat weight gradient inside the whole step, same box
for i in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export BP_NOWFLAT_S2=1; else unset BP_NOWFLAT_S2; fi
    python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-paint 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nowflat_s2=$v', d['value'], d['ms_per_step'])"
  done
done
