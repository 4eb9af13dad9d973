timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stem_on_the" 2>&1 | tail -2 | cut -c1-200
for i in 1 2; do
timeout 300 python tools/gpu_probe.py --model r50 --check --ops 2>&1 | grep -E "ms/iter|parity| stem"
done
