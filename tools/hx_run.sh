timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "small_batch_fully" 2>&1 | tail -4 | cut -c1-300
for i in 1 2; do
SMELTER_NO_FC_SMALL=1 timeout 300 python tools/gpu_probe.py --model r50 --ops 2>&1 | grep -E "ms/iter| FC "
timeout 300 python tools/gpu_probe.py --model r50 --check --ops 2>&1 | grep -E "ms/iter|parity| FC "
done
