#!/bin/bash
# the layers served by the 8-wave flattened-K kernels (conv_flat.hip), with and without them
L="0,32,64,4,2,1,64,256,256 1,64,32,4,2,1,64,128,128 0,16,8,7,1,3,64,512,512"
python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
BP_NOFLATG=1 BP_NOFLATW=1 BP_NOFLATH=1 python tools/conv_bench.py $L 2>&1 | grep -v amdgpu.ids
