for pp in 1 2 4; do
  echo "PPDMA=$pp"
  BP_IGEMM_PPDMA=$pp python tools/conv_bench.py 0,16,8,7,1,3,64,512,512 0,8,16,7,1,3,64,512,512 0,8,8,5,1,2,64,512,512 0,8,4,5,1,2,64,512,512
done
