for cfg in "BP_NOSTEM=1" "BP_SIDE_WGRAD=0" "BP_EPILOGUE_STATS=0" "X=1"; do
  echo "== $cfg"
  env $cfg python bench.py --legs none --steps 3 --warmup 1 > gpurun_out/np.json 2> gpurun_out/np.err && python -c "
import json; d=json.loads(open('gpurun_out/np.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || tail -3 gpurun_out/np.err
done
