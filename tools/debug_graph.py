import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, torch.nn.functional as F
from fastvim_amd.fastvim import PatchEmbed, FastVimT
from fastvim_amd.mamba_simple_faster import Mamba
from fastvim_amd.layernorm import rms_norm_fn
dev = "cuda"
def try_capture(name, fn):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capturing", name, flush=True)
    with torch.cuda.graph(g):
        out = fn()
    print("  captured", name, flush=True)
    g.replay(); torch.cuda.synchronize()
    print("  replayed", name, bool(torch.isfinite(out).all()), flush=True)

which = sys.argv[1]
if which == "pe":
    pe = PatchEmbed(224, 16, 3, 192, strict_img_size=False, dynamic_img_pad=True).to(dev)
    x = torch.randn(128, 3, 224, 224, device=dev)
    def f():
        pe.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = pe(x)
        y.float().square().mean().backward()
        return pe.proj.weight.grad
    try_capture("patch_embed", f)
elif which == "mixer":
    m = Mamba(192, token_size=[14, 14]).to(dev)
    h = torch.randn(128, 196, 192, device=dev, requires_grad=True)
    def f():
        m.zero_grad(); h.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(h)
        y.float().square().mean().backward()
        return h.grad
    try_capture("mixer", f)
elif which == "norm":
    w = torch.ones(192, device=dev, requires_grad=True)
    h = torch.randn(128, 196, 192, device=dev, requires_grad=True)
    r = torch.randn(128, 196, 192, device=dev, requires_grad=True)
    def f():
        w.grad = None; h.grad = None; r.grad = None
        y, ro = rms_norm_fn(h, w, None, residual=r, prenorm=True, residual_in_fp32=True)
        (y.square().mean() + ro.square().mean()).backward()
        return h.grad
    try_capture("norm", f)
if which == "model":
    from fastvim_amd.fastvim import VisionMamba
    depth = int(sys.argv[2]); bs = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    m = VisionMamba(img_size=224, depth=depth, embed_dim=192, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                    final_pool_type="mean", drop_path_rate=float(sys.argv[4]) if len(sys.argv) > 4 else 0.0).to(dev).train()
    x = torch.randn(bs, 3, 224, 224, device=dev)
    def f():
        m.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        y.float().square().mean().backward()
        return m.pos_embed.grad
    try_capture(f"model depth {depth} bs {bs}", f)
if which == "flat":
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.ddp import FlatGradAllReduce
    depth = int(sys.argv[2]); mode = sys.argv[3]
    m = VisionMamba(img_size=224, depth=depth, embed_dim=192, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                    final_pool_type="mean", drop_path_rate=0.05).to(dev).train()
    x = torch.randn(128, 3, 224, 224, device=dev)
    flat = FlatGradAllReduce(m.parameters()) if "flat" in mode else None
    def f():
        if flat is not None: flat.zero_()
        else: m.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        y.float().square().mean().backward()
        return m.pos_embed.grad
    if "eager" in mode:
        for _ in range(3): f()
        torch.cuda.synchronize()
    try_capture(f"model depth {depth} mode {mode}", f)
if which == "cmp":
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.ddp import FlatGradAllReduce
    depth = int(sys.argv[2])
    torch.manual_seed(0)
    m = VisionMamba(img_size=224, depth=depth, embed_dim=192, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                    final_pool_type="mean", drop_path_rate=0.0).to(dev).train()
    x = torch.randn(128, 3, 224, 224, device=dev)
    tgt = torch.softmax(torch.randn(128, 1000, device=dev), -1)
    flat = FlatGradAllReduce(m.parameters())
    def f():
        flat.zero_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        loss = torch.sum(-tgt * F.log_softmax(y.float(), dim=-1), dim=-1).mean()
        loss.backward()
        return loss.detach()
    l0 = f(); torch.cuda.synchronize()
    ref = flat.flat.clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        f(); f()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    print("eager repeat equal:", torch.equal(ref, flat.flat), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        lb = f()
    for it in range(3):
        g.replay(); torch.cuda.synchronize()
        diff = (flat.flat - ref).abs()
        print("replay", it, "loss", float(lb), float(l0), "finite", bool(torch.isfinite(flat.flat).all()), "maxdiff", float(diff.max()), flush=True)
        if not torch.equal(flat.flat, ref):
            off = 0
            for n, p in m.named_parameters():
                k = p.numel()
                d_ = diff[off:off + k]
                if float(d_.max()) > 0 or not torch.isfinite(flat.flat[off:off+k]).all():
                    print("   differs:", n, float(d_.max()), flush=True)
                off += k
            break
