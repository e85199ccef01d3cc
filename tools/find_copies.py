"""Which ops of the training step launch device-to-device memcpys / torch elementwise kernels (profiling aid)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from torch.profiler import profile, ProfilerActivity
import torch.nn.functional as F
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("aten::") and e.name in ("aten::copy_", "aten::add_", "aten::add", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::mul", "aten::sum", "aten::mean"):
        shapes = str(e.input_shapes)[:80]
        st = [s for s in (e.stack or []) if "fastvim_amd" in s or "bench" in s][:2]
        cnt[(e.name, shapes, tuple(st))] += 1
for (name, shapes, st), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(c, name, shapes, " <- ", [s.split("/")[-1][:70] for s in st])
