A="tools/conv_bench_bf16.py 0 128 128 3 1 1 64 64 64 5"
echo nw8; python $A 2>&1 | grep -v amdgpu
echo nw4; BP_BF16_NONW8=1 python $A 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_bf16_ops.py -q 2>&1 | tail -3
