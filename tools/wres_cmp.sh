S="0,16,32,4,2,1,64,512,512 1,32,16,4,2,1,64,256,256 0,16,8,7,1,3,64,512,512"
echo wres; python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
echo nowres; BP_IGEMM_NOWRES=1 python tools/conv_bench.py $S 2>&1 | grep -v amdgpu
