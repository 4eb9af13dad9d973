python -m pytest tests/test_gpu_ops.py tests/test_gpu_graph.py tests/test_gpu_full.py -m gpu -x -q -k "mbconv or mobilenet or mbv2" 2>&1 | cut -c1-300 | tail -3
python tools/gpu_probe.py --model mbv2 --ops --check > gpurun_out/r02t_probe_mbv2.log 2>&1; grep -E "ms/iter|parity|same launch" gpurun_out/r02t_probe_mbv2.log | cut -c1-120
python bench.py --model mbv2 > gpurun_out/r02t_bench_mbv2.json 2>gpurun_out/r02t_bench_mbv2.err; cut -c1-300 gpurun_out/r02t_bench_mbv2.json
