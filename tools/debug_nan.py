import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, torch.nn.functional as F
import bench
from fastvim_amd.ddp import FlatGradAllReduce
torch.manual_seed(1234)
dev = "cuda"
model = bench.build_model("T", 224, 0.05).to(dev).train()
gen = torch.Generator().manual_seed(100)
x = torch.randn(128, 3, 224, 224, generator=gen).to(dev)
tgt = bench.soft_targets(128, 1000, gen, dev)
flat = FlatGradAllReduce(model.parameters())
opt = torch.optim.AdamW(bench.param_groups(model, 0.05), lr=1e-3, fused=True, capturable=True)
for it in range(6):
    flat.zero_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = model(x)
    loss = torch.sum(-tgt * F.log_softmax(logits.float(), dim=-1), dim=-1).mean()
    loss.backward()
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
    print(it, float(loss), "nonfinite grads:", bad[:8], len(bad), "gradnorm", float(flat.flat.norm()))
    if bad: break
    opt.step()
    badp = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    if badp:
        print("nonfinite params", badp[:8]); break

print("---- graph mode")
def fwd_bwd():
    flat.zero_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = model(x)
    loss = torch.sum(-tgt * F.log_softmax(logits.float(), dim=-1), dim=-1).mean()
    loss.backward()
    return loss.detach()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lb = fwd_bwd()
for it in range(3):
    g.replay(); torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
    print(it, float(lb), "nonfinite grads:", bad[:12], len(bad))
