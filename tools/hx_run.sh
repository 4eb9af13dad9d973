for i in 1 2 3; do
SMELTER_XHEAD=0 timeout 300 python tools/gpu_probe.py --model r50 2>&1 | grep -E "ms/iter"
timeout 300 python tools/gpu_probe.py --model r50 --ops 2>&1 | grep -E "ms/iter|same launch"
done
