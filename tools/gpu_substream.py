"""experiment: run the batch as S independent sub-batches on S HIP streams (layers of different sub-batches overlap)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from smelter_amd import zoo  # noqa: E402
from smelter_amd.graph import Configuration, ONNXGraph  # noqa: E402

m = zoo.resnet50()
B = 64
x = torch.from_numpy(zoo.synthetic_images(B)).cuda()
for S in (1, 2, 4):
    g = ONNXGraph(m, Configuration(autotune=True))
    arena = torch.from_numpy(g.packWeights(B // S)).cuda()
    exs = [g.hipGraph(0, B // S, weight_arena=arena, upload_weights=False) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    xs = [x[i * (B // S):(i + 1) * (B // S)].contiguous() for i in range(S)]
    torch.cuda.synchronize()

    def step():
        for e, s, xi in zip(exs, streams, xs):
            e.encode(xi, s)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 30
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print("sub-batches=%d x %d images: %.3f ms/step  %.0f img/s" % (S, B // S, dt * 1e3, B / dt))
