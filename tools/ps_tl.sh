for cold in 0 1; do
if [ $cold = 1 ]; then export SM_TIME_COLD=1; fi
for sh in "64,14,14,256,1024,1,1,0" "64,14,14,256,1024,1,1,0 --res" "64,14,14,1024,256,1,1,0" "64,7,7,2048,512,1,1,0"; do echo "== $sh cold=$cold"; python tools/conv_bench.py --shape $sh --iters 10 --cfg 6,81,84,83,86 2>&1 | grep -E "^cfg|timeline" | cut -c1-400; done
done
