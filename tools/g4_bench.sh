#!/bin/bash
python tools/conv_bench.py 0,16,8,7,1,3,64,512,512 2>&1 | grep -v amdgpu.ids
BP_NOFLATH=1 python tools/conv_bench.py 0,16,8,7,1,3,64,512,512 2>&1 | grep -v amdgpu.ids
