for m in 0 fwd bwd 1; do
  BP_EPILOGUE_STATS=$m python bench.py --legs none --steps 10 --warmup 3 > gpurun_out/es_$m.json 2> gpurun_out/es_$m.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/es_$m.json').read().strip().splitlines()[-1]); print('$m', d['value'], d['ms_per_step'], d['roofline']['all_conv_kernels_ms_per_step'])"
done
