"""Which CPU ops of the training step issue device-to-device memcpys (profiling aid): chrome trace of one eager
step, every hipMemcpyAsync matched to the innermost enclosing aten / autograd op."""
import json, os, sys, tempfile, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
from fastvim_amd import fastvim as fv
from fastvim_amd.flat import FlatAdamW, FlatTrainingState
torch.manual_seed(0)
m = fv.FastVimT(img_size=224, drop_path_rate=0.05).cuda().train()
flat = FlatTrainingState(m)
opt = FlatAdamW(flat, m, lr=1e-3, weight_decay=0.05, no_decay=set(), ema_decay=0.9999)
x = torch.randn(16, 3, 224, 224, device="cuda")
tgt = torch.softmax(torch.randn(16, 1000, device="cuda"), -1)
def step():
    flat.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x)
    loss = torch.sum(-tgt * F.log_softmax(logits.float(), dim=-1), dim=-1).mean()
    loss.backward()
    flat.finish_backward()
    opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
path = os.path.join(tempfile.gettempdir(), "step_trace.json")
prof.export_chrome_trace(path)
ev = json.load(open(path))["traceEvents"]
cpu = [e for e in ev if e.get("cat") in ("cpu_op", "user_annotation", "python_function") and "dur" in e]
rt = [e for e in ev if e.get("cat") == "cuda_runtime" and "emcpy" in e.get("name", "")]
cnt = collections.Counter()
for r in rt:
    t = r["ts"]
    enc = [c for c in cpu if c["ts"] <= t <= c["ts"] + c["dur"] and c.get("tid") == r.get("tid")]
    enc.sort(key=lambda c: c["dur"])
    names = tuple(c["name"] for c in enc[:3])
    cnt[(r["name"], names)] += 1
for (name, names), c in cnt.most_common(20):
    print(c, name, "<-", names)
