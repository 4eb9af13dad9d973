timeout 900 python -m pytest tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do
SMELTER_MBCONV_STEM=0 timeout 300 python tools/gpu_probe.py --model mbv2 2>&1 | grep -E "ms/iter"
timeout 300 python tools/gpu_probe.py --model mbv2 --check --ops 2>&1 | grep -E "ms/iter|parity|stem"
done
