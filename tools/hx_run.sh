timeout 600 python -m pytest tests/test_gpu_graph.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-300
for i in 1 2 3; do
SMELTER_XHEAD_PROJ=0 timeout 300 python tools/gpu_probe.py --model r50 2>&1 | grep -E "ms/iter"
timeout 300 python tools/gpu_probe.py --model r50 --ops --check 2>&1 | grep -E "ms/iter|same launch|parity"
done
