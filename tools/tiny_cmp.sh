python tools/conv_bench.py 0,1,8,4,2,1,64,512,512 0,2,8,4,2,1,64,512,512
S="0,64,2,5,1,2,64,16,16 0,32,2,5,1,2,64,16,16 0,16,8,7,1,3,64,512,512"
echo pp; python tools/conv_bench.py $S
echo nopp; BP_IGEMM_PPCIN=8 python tools/conv_bench.py $S
