: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd $GRAFT_REPO_ROOT
S="0,8,1,5,1,2,64,512,512 0,1,1,3,1,1,64,512,512 0,2,8,4,2,1,64,512,512 0,8,16,8,4,2,64,256,256"
for v in "BP_WS_DEBUG=0" "BP_WS_DEBUG=1" "BP_WS_DEBUG=2" "BP_WS_DEBUG=3" "BP_WS_NSPLIT=2048" "BP_WS_NSPLIT=512" "BP_WS_NSPLIT=4096"; do
  echo "== $v"; env $v python tools/conv_bench.py $S 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad/wgrad/'
done
