python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "persistent_pointwise or (test_conv2d and (87 or 88))" 2>&1 | cut -c1-300 | tail -5 > gpurun_out/r02n_ps_tests.log; cat gpurun_out/r02n_ps_tests.log
for cold in 0 1; do
if [ $cold = 1 ]; then export SM_TIME_COLD=1; fi
for sh in "64,14,14,256,1024,1,1,0 --res" "64,14,14,256,1024,1,1,0" "64,14,14,1024,256,1,1,0" "64,28,28,512,1024,1,2,0 --res" "64,14,14,1024,512,1,1,0" "64,14,14,256,256,3,1,1" "64,28,28,128,512,1,1,0 --res"; do echo "== $sh cold=$cold"; python tools/conv_bench.py --shape $sh --iters 10 2>&1 | grep cfg | sort -k3 -n | awk '{ if (NR<=2 || $2>=81) print }'; done
done > gpurun_out/r02n_ps_bench.log 2>&1; grep -E "==|cfg 8[4-8]|cfg [0-7]" gpurun_out/r02n_ps_bench.log
