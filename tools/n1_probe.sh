for cfg in "BP_IGEMM_NOWRES=1" "BP_NOSTEM=1" "BP_NOTINY=1" "BP_EPILOGUE_STATS=0"; do
  echo "== $cfg"
  env $cfg python -m pytest tests/test_gpu_model.py -q -k "batch_sizes and 1-1" 2>&1 | grep "^E       Assertion\|passed\|failed" | cut -c1-250
done
