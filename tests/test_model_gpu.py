"""GPU parity of the FastVim backbone (fastvim_amd.fastvim.VisionMamba on the HIP kernels) against
golden vectors captured from the reference models/fastvim.py and against the fp64 oracle."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def _tiny(img, **kw):
    from fastvim_amd.fastvim import VisionMamba
    return VisionMamba(img_size=img, patch_size=16, depth=4, embed_dim=32, channels=3, num_classes=10,
                       rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                       if_abs_pos_embed=True, **kw)


@pytest.mark.parametrize("case", ["tiny_64x64", "tiny_48x80"])
def test_tiny_model_vs_reference_golden(case):
    c = load_golden("model_tiny.pt")[case]
    m = _tiny(c["img"], drop_path_rate=0.0).cuda().eval()
    m.load_state_dict(c["state_dict"], strict=True)
    hid = []
    hooks = [l.register_forward_hook(lambda mod, i, o: hid.append(o[0].detach())) for l in m.layers]
    logits = m(c["x"].cuda())
    for h in hooks:
        h.remove()
    assert _err(logits, c["logits"]) <= 2e-5 * max(1.0, c["logits"].abs().max().item()), _err(logits, c["logits"])
    for h, href in zip(hid, c["hiddens"]):
        assert _err(h, href) <= 2e-5 * max(1.0, href.abs().max().item())
    logits.backward(c["g"].cuda())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 2e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


def test_colwise_model_vs_reference_golden():
    """scanpath_type="colwise" (Pool_row in the paper, models/fastvim.py:45-51, 97-98): the patch grid is transposed
    before flattening; logits and gradients against the golden captured from the reference."""
    c = load_golden("model_colwise.pt")
    m = _tiny(c["img"], drop_path_rate=0.0, scanpath_type="colwise").cuda().eval()
    m.load_state_dict(c["state_dict"], strict=True)
    logits = m(c["x"].cuda())
    assert _err(logits, c["logits"]) <= 2e-5 * max(1.0, c["logits"].abs().max().item()), _err(logits, c["logits"])
    logits.backward(c["g"].cuda())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 2e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


def test_fastvim_t_vs_reference_golden_fp32():
    """BASELINE config 1 input (FastVim-T 224x224 bs=2, seeded weights) on the GPU, fp32."""
    from fastvim_amd.fastvim import FastVimT
    from oracle import make_state_dict
    c = load_golden("model_fastvim_t.pt")
    m = FastVimT(drop_path_rate=0.0).cuda().eval()
    m.load_state_dict(make_state_dict(seed=c["param_seed"], embed_dim=192, depth=24), strict=True)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(c["x_seed"]))
    assert torch.equal(x[0, 0, :2, :8], c["x_probe"])
    logits = m(x.cuda())
    s = max(1.0, c["logits"].abs().max().item())
    assert _err(logits, c["logits"]) <= 1e-4 * s, (_err(logits, c["logits"]), s)
    g = torch.randn(logits.shape, generator=torch.Generator().manual_seed(c["g_seed"]))
    logits.backward(g.cuda())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 1e-3 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


def test_fastvim_t_bf16_autocast_close_to_fp32_reference():
    from fastvim_amd.fastvim import FastVimT
    from oracle import make_state_dict
    c = load_golden("model_fastvim_t.pt")
    m = FastVimT(drop_path_rate=0.0).cuda().eval()
    m.load_state_dict(make_state_dict(seed=c["param_seed"], embed_dim=192, depth=24), strict=True)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(c["x_seed"]))
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        logits = m(x.cuda())
    ref = c["logits"]
    rel = (logits.float().cpu() - ref).norm() / ref.norm()
    assert rel <= 1.5e-2, rel   # 24 blocks of bf16 activations; fp32 residual stream keeps it under 1e-2 (measured 8.6e-3)


def test_fastvim_t_bf16_logits_vs_storage_rounded_oracle():
    """bf16 at the MODEL level against the fp64 oracle with a bf16 round trip at every tensor the HIP path stores in bf16 under
    autocast (patches, shadow weights, the patch projection's product, every block's normalised rows and mixer tensors, the
    final norm, the pooled feature, the head's logits -- ``fastvim_forward_oracle(storage_dtype=torch.bfloat16)``; the
    residual stream stays unrounded as in the HIP path).  At the mixer level that oracle pins the kernels to 2 ulps
    (tests/test_config34_gpu.py); over 24 blocks the occasional one-ulp flip of an intermediate is carried and amplified
    along the residual stream, so the model-level agreement is set by that, not by where the roundings sit: measured
    7.2e-3 relative L2 / 2 ulps of the logit scale against this oracle and 8.6e-3 against the reference's fp32 logits.
    Bounds: 1.2e-2 and 4 ulps here (the round-2 bound on the fp32 comparison was 3e-2; it is 1.5e-2 now)."""
    from fastvim_amd.fastvim import FastVimT
    from oracle import fastvim_forward_oracle, make_state_dict
    c = load_golden("model_fastvim_t.pt")
    sd = make_state_dict(seed=c["param_seed"], embed_dim=192, depth=24)
    m = FastVimT(drop_path_rate=0.0).cuda().eval()
    m.load_state_dict(sd, strict=True)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(c["x_seed"]))
    with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
        logits = m(x.cuda())
    ref = fastvim_forward_oracle(sd, x, compute_dtype=F64, storage_dtype=torch.bfloat16)
    rel = ((logits.double().cpu() - ref).norm() / ref.norm()).item()
    err, scale = _err(logits, ref), ref.abs().max().item()
    assert rel <= 1.2e-2, rel
    assert err <= 4 * 2.0 ** -8 * scale, (err, scale)


def test_image_gradient_flows_through_patch_embed():
    """An input image that requires grad gets its gradient through the fused patch projection (bf16 autocast: the
    projection + bias + position table GEMM; fp32: LinearFn): both must agree with each other to bf16 accuracy and the
    fp32 one with a finite-difference probe along a random direction."""
    torch.manual_seed(0)
    m = _tiny(64).cuda().eval()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    grads = {}
    for name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        xi = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=dt, enabled=dt != torch.float32):
            out = m(xi)
        out.float().square().sum().backward()
        assert xi.grad is not None and torch.isfinite(xi.grad).all() and xi.grad.abs().max() > 0
        grads[name] = xi.grad.clone()
    rel = (grads["bf16"] - grads["fp32"]).norm() / grads["fp32"].norm()
    assert rel <= 5e-2, rel
    d = torch.randn_like(x)
    d /= d.norm()
    f = lambda t: m(t).double().square().sum().item()
    with torch.no_grad():
        fd = (f(x + 1e-2 * d) - f(x - 1e-2 * d)) / 2e-2
    an = (grads["fp32"].double() * d.double()).sum().item()
    assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), (fd, an)


def test_training_mode_droppath_vs_oracle():
    """train() with stochastic depth: same per-sample scales fed to the model and the oracle."""
    from fastvim_amd import fastvim as fv
    from oracle import fastvim_forward_oracle
    torch.manual_seed(0)
    m = _tiny((64, 64), drop_path_rate=0.5).cuda().train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = torch.randn(4, 3, 64, 64)
    # entry i feeds block i (None where the block has no DropPath: layers 0 and 1, rate 0), entry 4 norm_f
    scales = [None, None] + [torch.tensor([0.0, 2.0, 2.0, 0.0]).roll(i) for i in range(3)]
    mods = [m.layers[2].drop_path, m.layers[3].drop_path, m.drop_path]
    assert all(isinstance(d, fv.DropPath) for d in mods) and isinstance(m.layers[1].drop_path, torch.nn.Identity)
    table = {id(d): s for d, s in zip(mods, scales[2:])}
    orig = fv.DropPath.row_scale
    fv.DropPath.row_scale = lambda self, t: table[id(self)].to(t.device)
    try:
        logits = m(x.cuda())
    finally:
        fv.DropPath.row_scale = orig
    sdc = {k: v.clone().requires_grad_() for k, v in sd.items()}
    ref = fastvim_forward_oracle(sdc, x, patch_size=16, depth=4, row_scales=scales, compute_dtype=F64)
    assert _err(logits, ref) <= 2e-5 * max(1.0, ref.abs().max().item()), _err(logits, ref)


def test_full_size_step_finite_and_deterministic():
    """BASELINE config 2: FastVim-T bs=128 bf16 fwd+bwd -- finite, bitwise reproducible."""
    from fastvim_amd.fastvim import FastVimT
    torch.manual_seed(0)
    m = FastVimT(drop_path_rate=0.0).cuda().train()
    x = torch.randn(128, 3, 224, 224, device="cuda")
    y = torch.randint(0, 1000, (128,), device="cuda")
    res = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(m(x).float(), y)
        loss.backward()
        res.append((loss.detach().clone(), m.layers[3].mixer.A_log.grad.clone(), m.pos_embed.grad.clone()))
    assert all(torch.isfinite(t).all() for t in res[0])
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("amp", [False, True])
def test_flat_training_state_matches_autograd(amp):
    """FlatTrainingState (flat params/grads, bf16 shadows, kernels accumulating straight into .grad)
    must produce the same gradients as plain autograd accumulation, twice in a row (zero + accumulate)."""
    import copy
    from fastvim_amd.flat import FlatTrainingState
    torch.manual_seed(0)
    m1 = _tiny((64, 64), drop_path_rate=0.0).cuda().train()
    with torch.no_grad():
        for n, p in m1.named_parameters():
            if n.endswith(("D", "D_b", "layernorm.bias", "conv1d.bias")):
                p.add_(0.1 * torch.randn_like(p))
    m2 = copy.deepcopy(m1)
    flat = FlatTrainingState(m2)
    x = torch.randn(4, 3, 64, 64, device="cuda")
    g = torch.randn(4, 10, device="cuda")

    def run(m):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            y = m(x)
        (y.float() * g).sum().backward()
        return y

    for it in range(2):
        m1.zero_grad(set_to_none=True)
        flat.zero_grad()
        y1, y2 = run(m1), run(m2)
        flat.finish_backward()
        assert torch.equal(y1, y2)
        p2 = dict(m2.named_parameters())
        for n, p in m1.named_parameters():
            a, b = p.grad, p2[n].grad
            assert b.data_ptr() >= flat.grad_flat.data_ptr() and b.data_ptr() < flat.grad_flat.data_ptr() + flat.grad_flat.numel() * 4
            tol = 1e-6 * max(1.0, a.abs().max().item())
            assert (a - b).abs().max().item() <= tol, (n, it, (a - b).abs().max().item())
    # an optimizer step through the flat views + shadow refresh keeps the two models in lock-step
    o1 = torch.optim.AdamW(m1.parameters(), lr=1e-2, fused=True)
    o2 = torch.optim.AdamW(m2.parameters(), lr=1e-2, fused=True)
    o1.step(); o2.step(); flat.refresh_shadow()
    from fastvim_amd.mixer_ops import defer_reductions
    from fastvim_amd.mamba_simple_faster import _SideStream
    defer_reductions(False)
    _SideStream.enabled = False
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        assert torch.equal(m1(x), m2(x))


def test_flat_adamw_matches_torch_adamw():
    """fv_adamw_flat == torch.optim.AdamW (two param groups) for several steps, + EMA + bf16 shadow."""
    import copy
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    torch.manual_seed(0)
    m1 = _tiny((64, 64), drop_path_rate=0.0).cuda().train()
    m2 = copy.deepcopy(m1)
    flat = FlatTrainingState(m2)
    no_decay = {n for n, p in m2.named_parameters() if p.ndim <= 1 or n.endswith(".bias") or n == "pos_embed"
                or getattr(p, "_no_weight_decay", False)}
    dec = [p for n, p in m1.named_parameters() if n not in no_decay]
    nod = [p for n, p in m1.named_parameters() if n in no_decay]
    o1 = torch.optim.AdamW([{"params": dec, "weight_decay": 0.05}, {"params": nod, "weight_decay": 0.0}], lr=3e-3)
    o2 = FlatAdamW(flat, m2, lr=3e-3, weight_decay=0.05, no_decay=no_decay, ema_decay=0.9)
    ema_ref = {n: p.detach().clone() for n, p in m1.named_parameters()}
    x = torch.randn(4, 3, 64, 64, device="cuda")
    for it in range(4):
        for m, zero in ((m1, lambda: m1.zero_grad(set_to_none=True)), (m2, flat.zero_grad)):
            zero()
            m(x).float().square().mean().backward()
        flat.finish_backward()
        if it == 2:
            for g in o1.param_groups:
                g["lr"] = 1e-3
            o2.set_lr(1e-3)
        o1.step(); o2.step()
        for n, p in m1.named_parameters():
            ema_ref[n].mul_(0.9).add_(p.detach(), alpha=0.1)
    p2 = dict(m2.named_parameters())
    for n, p in m1.named_parameters():
        assert (p - p2[n]).abs().max().item() <= 2e-6 * max(1.0, p.abs().max().item()), n
        assert (p2[n]._fv_shadow.float() - p2[n]).abs().max().item() <= 2.0 ** -8 * max(1e-3, p2[n].abs().max().item()), n
        off = flat.offsets[n]
        e = o2.ema[off:off + p.numel()].view_as(p)
        assert (e - ema_ref[n]).abs().max().item() <= 2e-6 * max(1.0, p.abs().max().item()), n
    from fastvim_amd.mixer_ops import defer_reductions
    defer_reductions(False)


def test_mm_fastvim_multiscale_features():
    """MM_FastVim.forward (models/fastvim.py:682-690): (B, C, H, W) maps of the LayerNorm-ed hidden states at
    out_indices equal what the oracle's block stack produces."""
    from fastvim_amd.fastvim import MM_FastVim
    from oracle import fastvim_forward_oracle
    torch.manual_seed(3)
    m = MM_FastVim(img_size=(64, 96), depth=4, embed_dim=32, out_indices=[1, 3], fused_add_norm=True,
                   residual_in_fp32=True, drop_path_rate=0.0).cuda().eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    sd.update({"head.weight": torch.zeros(1, 32), "head.bias": torch.zeros(1), "norm_f.weight": torch.ones(32)})
    x = torch.randn(2, 3, 64, 96)
    outs = m(x.cuda())
    assert len(outs) == 2 and outs[0].shape == (2, 32, 4, 6)
    _, hiddens = fastvim_forward_oracle(sd, x, patch_size=16, depth=4, compute_dtype=F64, return_hidden=True)
    for k, idx in enumerate((1, 3)):
        ref = torch.nn.functional.layer_norm(hiddens[idx].float(), (32,), sd[f"outnorm_{k}.weight"], sd[f"outnorm_{k}.bias"])
        ref = ref.view(2, 4, 6, 32).permute(0, 3, 1, 2)
        assert _err(outs[k], ref) <= 5e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("which", ["fastvim", "mae"])
def test_graph_replay_matches_eager_training(which):
    """The benchmarked step (fwd + loss + bwd + fused AdamW/EMA on the flat training state) replayed from a HIP graph
    must follow the eager trajectory: the kernels are deterministic and nothing in the step depends on the host.
    Bitwise-equal losses and parameters after 6 steps, at sizes where ROCm 7.2's default graph packet capture is
    known to replay correctly (DESIGN.md section 5 lists the configurations where it does not, and the switch
    bench.py sets for them)."""
    import copy
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    torch.manual_seed(0)
    if which == "fastvim":
        from fastvim_amd.fastvim import VisionMamba
        base = VisionMamba(img_size=224, depth=4, embed_dim=192, num_classes=100, rms_norm=True, residual_in_fp32=True,
                           fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.0).cuda()
        x = torch.randn(16, 3, 224, 224, device="cuda")
        tgt = torch.softmax(torch.randn(16, 100, device="cuda"), -1)

        def loss_of(model):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                logits = model(x)
            return torch.sum(-tgt * torch.log_softmax(logits.float(), -1), -1).mean()
    else:
        from fastvim_amd.models_mae import MaskedAutoencoderViM
        base = MaskedAutoencoderViM(img_size=224, depth=4, embed_dim=192, decoder_embed_dim=128, decoder_depth=1,
                                    rms_norm=True, residual_in_fp32=True, fused_add_norm=True).cuda()
        x = torch.randn(16, 3, 224, 224, device="cuda")
        noise = torch.rand(16, 196, device="cuda")

        def loss_of(model):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return model(x, noise=noise)[0]

    def make():
        m = copy.deepcopy(base).train()
        flat = FlatTrainingState(m)
        nd = {n for n, p in m.named_parameters() if p.ndim <= 1 or n.endswith(".bias") or n in m.no_weight_decay()
              or getattr(p, "_no_weight_decay", False)}
        return m, flat, FlatAdamW(flat, m, lr=1e-4, weight_decay=0.05, no_decay=nd, ema_decay=0.999)

    def one_step(m, flat, opt):
        flat.zero_grad()
        loss = loss_of(m)
        loss.backward()
        flat.finish_backward()
        opt.step()
        return loss.detach()

    m1, f1, o1 = make()
    eager = [one_step(m1, f1, o1).item() for _ in range(6)]
    assert all(l == l for l in eager) and eager[-1] < eager[0]
    m2, f2, o2 = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        warm = [one_step(m2, f2, o2).item() for _ in range(2)]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        lbuf = one_step(m2, f2, o2)
    replayed = []                       # capturing does not execute the step
    for _ in range(4):
        graph.replay()
        replayed.append(lbuf.item())
    assert warm + replayed == eager, (warm + replayed, eager)
    torch.cuda.synchronize()
    assert torch.equal(f1.param_flat, f2.param_flat)


def test_grouped_weight_gradients_match_plain_autograd():
    """At a size where the flat training state queues its weight gradients for the grouped launches (in_proj,
    out_proj, patch embed and x_proj -- the latter from bf16 dx_dbl, as the reference's autocast backward does), every
    parameter gradient must agree with plain autograd accumulation of the same model within bf16 rounding."""
    import copy
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatTrainingState
    from fastvim_amd.mixer_ops import defer_reductions
    from fastvim_amd.mamba_simple_faster import _GroupedWgrad, group_wgrads
    torch.manual_seed(1)
    m1 = VisionMamba(img_size=224, depth=2, embed_dim=192, num_classes=50, rms_norm=True, residual_in_fp32=True,
                     fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.0).cuda().train()
    m2 = copy.deepcopy(m1)
    flat = FlatTrainingState(m2)
    assert _GroupedWgrad.enabled
    x = torch.randn(32, 3, 224, 224, device="cuda")
    g = torch.randn(32, 50, device="cuda")
    for m in (m1, m2):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        (y.float() * g).sum().backward()
    assert len(_GroupedWgrad.jobs) == 2 * 4 + 1          # per block: in_proj, out_proj, x_proj x 2; + patch embed
    flat.finish_backward()
    torch.cuda.synchronize()
    p2 = dict(m2.named_parameters())
    for n, p in m1.named_parameters():
        a, b = p.grad.float(), p2[n].grad.float()
        tol = (2e-2 if "x_proj" in n else 2e-3) * max(1e-3, a.abs().max().item())
        assert (a - b).abs().max().item() <= tol, (n, (a - b).abs().max().item(), a.abs().max().item())
    defer_reductions(False)
    group_wgrads(False)


_GEMM_ENTRY_POINTS = ("bmm", "baddbmm", "matmul", "mm", "addmm", "einsum")


def _trap_library_gemms(mp, hits):
    import torch.nn.functional as F

    def trap(name):
        def f(*a, **k):
            hits.append(name)
            raise AssertionError(f"library GEMM {name} on the product path")
        return f

    for mod, name in ([(torch, n) for n in _GEMM_ENTRY_POINTS] + [(F, "linear"), (F, "conv2d"), (F, "conv3d"),
                      (torch.Tensor, "__matmul__"), (torch.Tensor, "matmul"), (torch.Tensor, "mm"), (torch.Tensor, "bmm"),
                      (torch.Tensor, "baddbmm_")]):
        mp.setattr(mod, name, trap(name))


@pytest.mark.parametrize("amp,d,batch,img", [(torch.bfloat16, 192, 64, 224), (torch.float32, 192, 16, 224),
                                             (torch.float32, 768, 4, 224), (torch.bfloat16, 768, 8, 224)])
def test_training_step_calls_no_library_gemm(monkeypatch, amp, d, batch, img):
    """The product path is hand-written HIP end to end, in bf16 AND in fp32 (the reference's default precision): with
    every torch matmul / linear / conv entry point turned into an error, a FastVim training step on the flat training
    state (patch embed, in/out/x projections forward, data and weight gradients, head, loss, optimizer) still runs.
    bf16: the tuned MFMA GEMMs, with the fp32-MFMA kernel for the shapes they do not take (the head's weight gradient at
    batch 8: 8 tokens); fp32: the fp32-MFMA kernel (csrc/gemm_f32.hip) everywhere -- so the fp32 golden tests pin this
    build's GEMM code, not rocBLAS."""
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    from fastvim_amd.losses import SoftTargetCrossEntropy
    torch.manual_seed(0)
    m = VisionMamba(img_size=img, depth=2, embed_dim=d, num_classes=1000, rms_norm=True, residual_in_fp32=True,
                    fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.1).cuda().train()
    x = torch.randn(batch, 3, img, img, device="cuda")
    tgt = torch.softmax(torch.randn(batch, 1000, device="cuda"), -1)
    crit = SoftTargetCrossEntropy()
    hits = []
    with FlatTrainingState(m) as flat:
        opt = FlatAdamW(flat, m, lr=1e-3, no_decay=set())
        with monkeypatch.context() as mp:
            _trap_library_gemms(mp, hits)
            flat.zero_grad()
            with torch.autocast("cuda", dtype=amp, enabled=amp != torch.float32):
                loss = crit(m(x), tgt)
            loss.backward()
            opt.step()
            assert not hits and torch.isfinite(loss)
            assert torch.isfinite(flat.grad_flat).all() and flat.grad_flat.abs().max() > 0


def test_eager_fp32_model_without_flat_state_calls_no_library_gemm(monkeypatch):
    """The same for a plain module (no flat training state: autograd-visible weight gradients, eager split-K sums) and
    for the un-pooled Vim baseline mixer."""
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.vim import VisionMamba as Vim
    hits = []
    torch.manual_seed(0)
    models = [VisionMamba(img_size=64, depth=2, embed_dim=64, num_classes=10, rms_norm=True, residual_in_fp32=True,
                          fused_add_norm=True, final_pool_type="mean").cuda().train(),
              Vim(img_size=64, patch_size=16, depth=2, embed_dim=64, channels=3, num_classes=10, rms_norm=True,
                  residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True,
                  if_cls_token=True, use_middle_cls_token=True).cuda().train()]
    with monkeypatch.context() as mp:
        _trap_library_gemms(mp, hits)
        for m in models:
            out = m(torch.randn(2, 3, 64, 64, device="cuda"))
            out.square().mean().backward()
            assert not hits and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize("which", ["channel", "mae", "vim"])
def test_other_model_families_call_no_library_gemm(monkeypatch, which):
    """The channel model (BASELINE config 5: per-channel patch embedding, channel-embedding epilogue, head), the MAE
    pre-training model (masked mixer, decoder embedding / prediction layers) and the Vim baseline under bf16 autocast on
    the flat training state: no torch matmul / linear / conv entry point is reached (round 2's config-5 trace still showed
    three Tensile kernels per step: the head's nn.Linear)."""
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    torch.manual_seed(0)
    if which == "channel":
        from fastvim_amd.models_channel_mamba_faster import VisionMamba as ChanVim
        m = ChanVim(img_size=64, patch_size=16, depth=2, embed_dim=64, channels=4, num_classes=10, rms_norm=True,
                    residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean", hcs=False)
        x = torch.randn(8, 4, 64, 64, device="cuda")
        loss_of = lambda out: out.float().square().mean()
    elif which == "mae":
        from fastvim_amd.models_mae import MaskedAutoencoderViM
        m = MaskedAutoencoderViM(img_size=64, patch_size=16, depth=2, embed_dim=64, decoder_embed_dim=32, decoder_depth=1,
                                 rms_norm=True, residual_in_fp32=True, fused_add_norm=True)
        x = torch.randn(8, 3, 64, 64, device="cuda")
        loss_of = lambda out: out[0]
    else:
        from fastvim_amd.vim import VisionMamba as Vim
        m = Vim(img_size=64, patch_size=16, depth=2, embed_dim=64, channels=3, num_classes=10, rms_norm=True,
                residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True,
                if_cls_token=True, use_middle_cls_token=True)
        x = torch.randn(8, 3, 64, 64, device="cuda")
        loss_of = lambda out: out.float().square().mean()
    m = m.cuda().train()
    hits = []
    with FlatTrainingState(m) as flat:
        opt = FlatAdamW(flat, m, lr=1e-3, no_decay=set())
        with monkeypatch.context() as mp:
            _trap_library_gemms(mp, hits)
            flat.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = m(x, mask_ratio=0.75) if which == "mae" else m(x)
            loss = loss_of(out)
            loss.backward()
            opt.step()
            assert not hits and torch.isfinite(loss)
