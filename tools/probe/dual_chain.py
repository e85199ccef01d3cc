"""Would two half-batch chains on two streams fill each other's ramps and tails?  Experiment: TWO copies of FastVim-T, each
with its own flat training state, each on its own stream with half the batch, captured into ONE HIP graph (parallel
branches), against one copy with the full batch.  Same total work except the second optimizer pass (46 us) and weight
gradients over half the tokens each.  usage: python tools/probe/dual_chain.py [batch] [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from fastvim_amd import fastvim as fv
from fastvim_amd.flat import FlatAdamW, FlatTrainingState
from fastvim_amd.losses import SoftTargetCrossEntropy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
crit = SoftTargetCrossEntropy()

def make():
    torch.manual_seed(1234)
    m = fv.FastVimT(img_size=224, drop_path_rate=0.05).to(dev).train()
    flat = FlatTrainingState(m)
    nd = {n for n, p in m.named_parameters() if p.ndim <= 1 or n.endswith(".bias") or n in m.no_weight_decay() or getattr(p, "_no_weight_decay", False)}
    return m, flat, FlatAdamW(flat, m, lr=1e-3, weight_decay=0.05, no_decay=nd, ema_decay=0.9999)

def fwd_bwd(m, x, t):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x)
    loss = crit(logits, t)
    loss.backward()
    return loss.detach()

def timeit(step_fn):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step_fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step_fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out

gen = torch.Generator().manual_seed(0)
x = torch.randn(B, 3, 224, 224, generator=gen).to(dev)
tgt = torch.softmax(torch.randn(B, 1000, generator=gen), -1).to(dev)

# (A) one chain, full batch
m, flat, opt = make()
def single():
    flat.zero_grad()
    l = fwd_bwd(m, x, tgt)
    flat.finish_backward(); opt.step()
    return l
ms_a, la = timeit(single)
print(f"one chain, batch {B}: {ms_a:.3f} ms/step  loss {la.item():.4f}")
flat.close(); del m, flat, opt

# (B) two chains, half batch each, two streams inside one graph
m1, f1, o1 = make(); m2, f2, o2 = make()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h = B // 2
x1, x2, t1, t2 = x[:h].contiguous(), x[h:].contiguous(), tgt[:h].contiguous(), tgt[h:].contiguous()
def dual():
    cur = torch.cuda.current_stream()
    f1.grad_flat.zero_(); f2.grad_flat.zero_()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        l1 = fwd_bwd(m1, x1, t1)
    with torch.cuda.stream(s2):
        l2 = fwd_bwd(m2, x2, t2)
    cur.wait_stream(s1); cur.wait_stream(s2)
    f1.finish_backward()            # the queues are process-wide: both chains' weight gradients and partial sums
    o1.step(); o2.step()
    return l1 + l2
ms_b, lb = timeit(dual)
print(f"two chains of batch {h} on two streams: {ms_b:.3f} ms/step ({B} images)  loss sum {lb.item():.4f}")

# (C) the same two chains on ONE stream (what splitting alone costs)
def serial2():
    f1.grad_flat.zero_(); f2.grad_flat.zero_()
    l1 = fwd_bwd(m1, x1, t1); l2 = fwd_bwd(m2, x2, t2)
    f1.finish_backward(); o1.step(); o2.step()
    return l1 + l2
ms_c, lc = timeit(serial2)
print(f"two chains of batch {h} on one stream : {ms_c:.3f} ms/step")
