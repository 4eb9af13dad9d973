python -m pytest tests/test_gpu_ops.py tests/test_gpu_graph.py -m gpu -x -q -k "mbconv or mobilenet or mbv2" 2>&1 | cut -c1-300 | tail -4
python tools/gpu_probe.py --model mbv2 --ops --check > gpurun_out/r02r_probe_mbv2.log 2>&1; grep -E "ms/iter|parity|same launch" gpurun_out/r02r_probe_mbv2.log | cut -c1-260
SMELTER_MBCONV=0 python tools/gpu_probe.py --model mbv2 2>&1 | grep -E "ms/iter"
