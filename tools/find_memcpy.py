"""Which host ops issue the device-to-device memcpy nodes (__amd_rocclr_copyBuffer) of one eager training step?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
from bench import build_model, soft_targets
from fastvim_amd.flat import FlatAdamW, FlatTrainingState
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = build_model("T", 224, 0.05).to(dev).train()
gen = torch.Generator().manual_seed(1)
x = torch.randn(32, 3, 224, 224, generator=gen).to(dev)
tgt = soft_targets(32, 1000, gen, dev)
flat = FlatTrainingState(model)
opt = FlatAdamW(flat, model, lr=1e-3, weight_decay=0.05, no_decay=set(), ema_decay=0.9999)
def step():
    flat.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = model(x)
    loss = torch.sum(-tgt * F.log_softmax(logits.float(), dim=-1), dim=-1).mean()
    loss.backward()
    flat.finish_backward()
    opt.step()
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()
n = 0
for e in evs:
    if "emcpy" in e.name or "copyBuffer" in e.name:
        n += 1
        # walk up the CPU parents
        chain, p = [], e.cpu_parent if hasattr(e, "cpu_parent") else None
        while p is not None and len(chain) < 6:
            chain.append(p.name); p = p.cpu_parent
        print(n, e.name, "device_time", getattr(e, "device_time", None), "<-", " <- ".join(chain), flush=True)
        if e.stack:
            print("    ", " | ".join(s for s in e.stack[:6]))
print("cpu ops that launched memcpy:")
for e in evs:
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy") and e.device_type == torch.autograd.DeviceType.CPU:
        for k in e.kernels:
            if "emcpy" in k.name or "copyBuffer" in k.name:
                st = " | ".join(e.stack[:5]) if e.stack else ""
                par = e.cpu_parent.name if e.cpu_parent is not None else ""
                print("  ", e.name, list(e.input_shapes) if e.input_shapes else "", "parent:", par, st)
