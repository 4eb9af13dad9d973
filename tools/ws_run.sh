python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "weight_stationary" > gpurun_out/r02b_ws_tests.log 2>&1; tail -3 gpurun_out/r02b_ws_tests.log | cut -c1-300
export SM_TIME_COLD=1
for sh in 64,56,56,64,64,3,1,1 16,512,512,64,32,3,1,1 16,512,512,32,32,3,1,1 16,256,256,64,64,3,1,1 16,256,256,32,64,3,1,1; do echo "== $sh"; python tools/conv_bench.py --shape $sh --iters 10 --cfg 61,62,63,64,65,66,67 2>&1 | grep cfg; done > gpurun_out/r02b_ws_bench.log 2>&1; cat gpurun_out/r02b_ws_bench.log
