for dom in "32,192,64" "64,384,96" "96,576,160" "160,960,320"; do
echo "domain $dom"; SMELTER_MBCONV_DOMAIN=$dom timeout 300 python tools/gpu_probe.py --model mbv2 --check 2>&1 | grep -E "ms/iter|parity"
done
SMELTER_MBCONV_DOMAIN="160,960,320" timeout 300 python tools/gpu_probe.py --model mbv2 --ops 2>&1 | grep -E "same launch"
