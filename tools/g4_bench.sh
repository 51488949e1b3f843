#!/bin/bash
python tools/conv_bench.py 0,32,64,4,2,1,64,256,256 1,64,32,4,2,1,64,128,128 2>&1 | grep -v amdgpu.ids
BP_NOFLATW=1 python tools/conv_bench.py 0,32,64,4,2,1,64,256,256 1,64,32,4,2,1,64,128,128 2>&1 | grep -v amdgpu.ids
