for d in 0 1 2 4 3 7 15; do echo -n "dbg=$d  "; BP_DMA_DBG=$d python tools/conv_bench.py 0,128,128,3,1,1,64,64,64 2>&1 | grep -v amdgpu | cut -c1-110; done
