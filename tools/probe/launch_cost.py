"""Per-node cost of dependent tiny kernels in a replayed HIP graph (what every kernel boundary of the step costs)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import fastvim_amd  # sets the graph packet-capture switch
import torch
x = torch.zeros(64, device="cuda"); big = torch.zeros(25088 * 768, device="cuda", dtype=torch.bfloat16)
def timeit(fn, n):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); 
        for _ in range(10): g.replay()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 / n * 1e3
print("tiny fill, 200 nodes: %.2f us/node" % timeit(lambda: x.fill_(1.0), 200))
print("tiny add (dependent), 200 nodes: %.2f us/node" % timeit(lambda: x.add_(1.0), 200))
print("38 MB fill: %.2f us/node" % timeit(lambda: big.fill_(1.0), 50))
