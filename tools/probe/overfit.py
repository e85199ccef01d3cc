"""End-to-end sanity of the training path: FastVim-T (bf16 autocast, flat training state, fused AdamW + EMA, HIP-graph replay)
on ONE fixed synthetic batch with hard labels -- the loss must fall towards zero (the model memorises 128 images).
usage: python tools/probe/overfit.py [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from fastvim_amd import fastvim as fv
from fastvim_amd.flat import FlatAdamW, FlatTrainingState
from fastvim_amd.losses import SoftTargetCrossEntropy
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
torch.manual_seed(0)
m = fv.FastVimT(img_size=224, drop_path_rate=0.0).cuda().train()
g = torch.Generator().manual_seed(1)
x = torch.randn(128, 3, 224, 224, generator=g).cuda()
y = torch.randint(0, 1000, (128,), generator=g)
tgt = torch.zeros(128, 1000).scatter_(1, y[:, None], 1.0).cuda()
flat = FlatTrainingState(m)
nd = {n for n, p in m.named_parameters() if p.ndim <= 1 or n.endswith(".bias") or n in m.no_weight_decay() or getattr(p, "_no_weight_decay", False)}
opt = FlatAdamW(flat, m, lr=5e-4, weight_decay=0.05, no_decay=nd, ema_decay=0.999)
crit = SoftTargetCrossEntropy()
def step():
    flat.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x)
    loss = crit(logits, tgt)
    loss.backward(); flat.finish_backward(); opt.step()
    return loss.detach()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    l0 = step().item(); step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    lb = step()
hist = [l0]
for i in range(steps):
    gr.replay()
    if (i + 1) % 100 == 0:
        hist.append(lb.item())
m.eval()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    acc = (m(x).argmax(-1).cpu() == y).float().mean().item()
print(f"FastVim-T, one fixed batch of 128, {steps} graph-replayed steps: loss {' -> '.join(f'{h:.4f}' for h in hist)}; train accuracy {acc:.3f}; "
      f"parameters finite: {bool(torch.isfinite(flat.param_flat).all())}")
