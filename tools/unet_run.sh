python -m pytest tests/test_gpu_graph.py tests/test_gpu_full.py -m gpu -x -q -k "unet" 2>&1 | cut -c1-300 | tail -4
python tools/gpu_probe.py --model unet --ops --check > gpurun_out/r02v_probe_unet.log 2>&1; grep -E "ms/iter|parity|same launch" gpurun_out/r02v_probe_unet.log | cut -c1-200
