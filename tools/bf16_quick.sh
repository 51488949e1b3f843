python -m pytest tests/test_gpu_bf16_ops.py -q 2>&1 | tail -2
for a in "0 128 128 3 1 1 64 64 64 5" "0 16 32 4 2 1 64 512 512 5" "1 32 16 4 2 1 64 256 256 5" "0 16 8 7 1 3 64 512 512 5" "0 64 128 4 2 1 64 128 128 5"; do echo "== $a"; python tools/conv_bench_bf16.py $a 2>&1 | grep -v amdgpu; done
