#!/bin/bash
run() { python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-paint $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  run auto
  BP_EPILOGUE_STATS=fwd run fwd_only
done
