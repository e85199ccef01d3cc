"""Does a process group's watchdog thread (it polls the events of finished collectives) break a graph capture on another
thread?  One rank over RCCL: a burst of async all-reduces, then -- while the watchdog is still retiring them -- capture a
graph, 40 times.  usage: python tools/probe/capture_vs_watchdog.py global|thread_local"""
import os, sys
import torch, torch.distributed as dist
mode = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
y = torch.zeros(1 << 16, device="cuda")
ok = 0
for it in range(40):
    works = [dist.all_reduce(x, async_op=True) for _ in range(20)]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
            for _ in range(200):
                y.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    for w in works:
        w.wait()
    g.replay()
    ok += 1
torch.cuda.synchronize()
print(f"capture_error_mode={mode}: {ok} captures beside a busy watchdog, no error; y[0] = {y[0].item()}")
dist.destroy_process_group()
