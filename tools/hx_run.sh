for i in 1 2 3; do
SMELTER_MBCONV_STEM=0 timeout 300 python tools/gpu_probe.py --model mbv2 2>&1 | grep -E "ms/iter"
timeout 300 python tools/gpu_probe.py --model mbv2 --check 2>&1 | grep -E "ms/iter|parity"
done
