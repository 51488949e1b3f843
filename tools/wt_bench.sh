#!/bin/bash
# the weight gradients served by wgrad_tiles_dma_kernel: trunk 128<->128 k3 and the wide stride-2 k4 layers
python tools/conv_bench.py 0,128,128,3,1,1,64,64,64 0,64,128,4,2,1,64,128,128 0,32,64,4,2,1,64,256,256 1,128,64,4,2,1,64,64,64 2>&1 | grep -v amdgpu.ids | sed 's/.*dgrad[^|]*| //'
