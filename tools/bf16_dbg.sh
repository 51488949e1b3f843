for d in 0 1 2 4 3 7 15; do echo -n "dbg=$d  "; BP_BF16_DBG=$d python tools/conv_bench_bf16.py 0 128 128 3 1 1 64 64 64 5 2>&1 | grep "forward"; done
