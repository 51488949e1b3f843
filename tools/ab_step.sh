#!/bin/bash
# Same-box A/B of the whole training step: bash tools/ab_step.sh "<ENV=1 ...>" ["<bench args>"]
# (box-to-box variation is ~1 %: only numbers of one call compare)
run() { env $1 python bench.py --legs none --steps 16 --warmup 3 --no-cpu-baseline --no-paint $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1]', d['value'], 'tiles/s', d['ms_per_step'], 'ms')"; }
for i in 1 2; do
  run "BP_AB_BASE=1" "$2"
  run "$1" "$2"
done
