#!/bin/bash
run() { env $1 python bench.py --legs none --steps 16 --warmup 3 --no-cpu-baseline --no-paint 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
run BP_X=1
for g in 384 448 640 768; do run BP_FLAT_GRID=$g; done
run BP_X=1
for g in 192 224 320; do run BP_FLATG_GRID=$g; done
for g in 448 640; do run BP_WRES_GRID=$g; done
