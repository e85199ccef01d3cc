import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from fastvim_amd import fastvim as fv
from fastvim_amd import mamba_simple_faster as msf
dev = "cuda"
x = torch.randn(4, 3, 224, 224, device=dev, generator=torch.Generator(device="cuda").manual_seed(5))
def run(chain, train, grad):
    torch.manual_seed(0)
    m = fv.FastVimT(img_size=224, drop_path_rate=0.0).to(dev)
    m.train(train)
    if not chain: m._chainable = lambda *a, **k: False
    rec = []
    orig = msf.FastVimMixerFn.forward
    def spy(ctx, hidden, *a, _o=orig, _r=rec):
        _r.append(hidden.detach().clone()); return _o(ctx, hidden, *a)
    msf.FastVimMixerFn.forward = staticmethod(spy)
    with torch.set_grad_enabled(grad), torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(x)
    msf.FastVimMixerFn.forward = staticmethod(orig)
    return logits.detach(), rec
ref = run(False, False, False)
for cfg in [(True, False, False), (True, True, False), (True, False, True), (True, True, True), (False, True, True)]:
    lg, rec = run(*cfg)
    first = next((i for i, (a, b) in enumerate(zip(rec, ref[1])) if not torch.equal(a, b)), None)
    print("chain=%s train=%s grad=%s" % cfg, "logits equal ref:", torch.equal(lg, ref[0]), "first differing mixer input:", first)
