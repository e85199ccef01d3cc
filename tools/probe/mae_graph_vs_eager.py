"""MAE pre-training step (FastVim-B encoder, bench.py --model M) at batch B with a FIXED masking noise: graph replays vs
the eager trajectory, bit for bit -- run once per DEBUG_CLR_GRAPH_PACKET_CAPTURE setting (the runtime reads it at start-up).
usage: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0|1 python tools/probe/mae_graph_vs_eager.py [batch] [steps]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd.flat import FlatAdamW, FlatTrainingState
from fastvim_amd.models_mae import mae_FastVim_base_dec512d2b
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
noise = torch.rand(B, 196, generator=torch.Generator().manual_seed(2)).cuda()

def make():
    torch.manual_seed(1234)
    m = mae_FastVim_base_dec512d2b(img_size=224).cuda().train()
    flat = FlatTrainingState(m)
    nd = {n for n, p in m.named_parameters() if p.ndim <= 1 or n.endswith(".bias") or n in m.no_weight_decay()
          or getattr(p, "_no_weight_decay", False)}
    return m, flat, FlatAdamW(flat, m, lr=1.5e-4 * B / 256, betas=(0.9, 0.95), weight_decay=0.05, no_decay=nd, ema_decay=0.9999)

def one_step(m, flat, opt):
    flat.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = m(x, noise=noise)[0]
    loss.backward()
    flat.finish_backward()
    opt.step()
    return loss.detach()

m1, f1, o1 = make()
eager = [one_step(m1, f1, o1).item() for _ in range(2 + steps)]
p_eager = f1.param_flat.clone()
f1.close()
m2, f2, o2 = make()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    warm = [one_step(m2, f2, o2).item() for _ in range(2)]
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lb = one_step(m2, f2, o2)
rep = []
for _ in range(steps):
    g.replay(); rep.append(lb.item())
torch.cuda.synchronize()
same = (warm + rep == eager)
first_bad = next((i for i, (a, b) in enumerate(zip(warm + rep, eager)) if a != b), None)
print(f"packet_capture={os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')} batch={B}: losses equal bit for bit: {same}"
      f" (first differing step: {first_bad}); params equal: {torch.equal(p_eager, f2.param_flat)};"
      f" max |dp| {(p_eager - f2.param_flat).abs().max().item():.3e}; last loss eager {eager[-1]!r} graph {rep[-1]!r}")
