# the two "proof" shapes of VERDICT r1 item 4 + the best configs on them (warm, back-to-back launches: a GEMM-benchmark convention)
for sh in 1024,14,14,256,256,3,1,1 64,128,128,4096,4096,1,1,0 16,128,128,256,256,3,1,1; do echo "== $sh"; timeout 300 python tools/conv_bench.py --shape $sh --iters 5 2>&1 | grep cfg | sort -k3 -n | head -4; done
