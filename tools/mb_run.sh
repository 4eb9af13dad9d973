python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "mbconv" 2>&1 | cut -c1-300 | tail -2
python tools/mbconv_bench.py 2>&1 | grep -v amdgpu
