import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd import fastvim as fv
from fastvim_amd import mamba_simple_faster as msf
dev = "cuda"
x = torch.randn(4, 3, 224, 224, device=dev, generator=torch.Generator(device="cuda").manual_seed(5))
outs = []
for chain in (True, False):
    torch.manual_seed(0)
    m = fv.FastVimT(img_size=224, drop_path_rate=0.0).to(dev).train()
    if not chain:
        m._chainable = lambda *a, **k: False
    rec = []
    orig = msf.FastVimMixerFn.forward
    def spy(ctx, hidden, *a, _o=orig, _r=rec):
        out = _o(ctx, hidden, *a)
        xz = [t for t in ctx.saved_tensors if t is not None and t.dim() == 3 and t.shape[-1] == 1536]
        _r.append((hidden.detach().clone(), xz[0].detach().clone() if xz else None, out.detach().clone()))
        return out
    msf.FastVimMixerFn.forward = staticmethod(spy)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x)
    msf.FastVimMixerFn.forward = staticmethod(orig)
    outs.append((logits, rec))
print("logits equal", torch.equal(outs[0][0], outs[1][0]))
for i, (a, b) in enumerate(zip(outs[0][1], outs[1][1])):
    eq_h = torch.equal(a[0], b[0])
    print(i, "mixer input eq", eq_h, "" if eq_h else (a[0].float() - b[0].float()).abs().max().item())
    if not eq_h:
        dd = (a[0].float() - b[0].float()).abs().view(-1, a[0].shape[-1])
        print("   rows", (dd.max(1).values > 0).nonzero().flatten()[:12].tolist(), "n rows", int((dd.max(1).values > 0).sum()))
        break
