#!/bin/bash
python tools/conv_bench.py 1,32,16,4,2,1,64,256,256 0,16,32,4,2,1,64,512,512 0,32,64,4,2,1,64,256,256 2>&1 | grep -v amdgpu.ids
BP_NOFLATG_THIN=1 python tools/conv_bench.py 1,32,16,4,2,1,64,256,256 0,16,32,4,2,1,64,512,512 2>&1 | grep -v amdgpu.ids
