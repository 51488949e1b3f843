for d in 0 1 2 3 7; do echo "dbg=$d"; BP_WRES_DBG=$d python tools/conv_bench.py 0,16,32,4,2,1,64,512,512 0,16,8,7,1,3,64,512,512 2>&1 | grep -v amdgpu | cut -c1-75; done
