python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "stem_on_the_weight" 2>&1 | cut -c1-300 | tail -8 > gpurun_out/r02m_stem_tests.log; cat gpurun_out/r02m_stem_tests.log
for m in r50 unet mbv2; do python tools/gpu_probe.py --model $m --ops --check > gpurun_out/r02m_probe_$m.log 2>&1; grep -E "ms/iter|parity|stem|wst" gpurun_out/r02m_probe_$m.log | cut -c1-250 | head -8; done
