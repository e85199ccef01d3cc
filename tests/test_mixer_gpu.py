"""GPU parity of the fused FastVim mixer module (HIP kernels through the C ABI) against the golden
vectors captured from the reference mixer (mamba_simple_faster.py) and against the fp64 oracle."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _sd(c):
    from oracle import make_state_dict
    if "state_dict" in c:
        return c["state_dict"]
    r = c["param_recipe"]
    full = make_state_dict(seed=r["seed"], embed_dim=r["embed_dim"], depth=r["depth"])
    return {k[len(r["prefix"]):]: v for k, v in full.items() if k.startswith(r["prefix"])}


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def _build(sd, token_size, **kw):
    from fastvim_amd.mamba_simple_faster import Mamba
    d_model = sd["in_proj.weight"].shape[1]
    m = Mamba(d_model, token_size=list(token_size), **kw).cuda()
    missing = m.load_state_dict(sd, strict=True)
    return m


@pytest.mark.parametrize("case", ["d32_4x4", "d32_3x5", "d192_14x14"])
@pytest.mark.parametrize("transposed", [False, True])
def test_mixer_fp32_vs_reference_golden(case, transposed):
    c = load_golden("mixer.pt")[case]
    rows, cols = c["token_size"]
    m = _build(_sd(c), (rows, cols))
    h = c["hidden"]
    Bsz, Ltok, d = h.shape
    perm = (lambda t: t.reshape(Bsz, rows, cols, -1).transpose(1, 2).reshape(Bsz, Ltok, -1)) if transposed else (lambda t: t)
    hg = perm(h).contiguous().cuda().requires_grad_()
    y = m(hg, transposed_grid=transposed)
    ref = perm(c["out"])
    assert _err(y, ref) <= 1e-5 * max(1.0, ref.abs().max().item()), _err(y, ref)
    y.backward(perm(c["g"]).contiguous().cuda())
    dref = perm(c["dhidden"])
    assert _err(hg.grad, dref) <= 2e-5 * max(1.0, dref.abs().max().item()), _err(hg.grad, dref)
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 1e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


@pytest.mark.parametrize("case", ["d32_3x5", "d192_14x14"])
def test_mixer_bf16_vs_fp64_oracle(case):
    """bf16 storage + fp32 math under autocast: compare with the fp64 oracle on the bf16-rounded
    input; tolerance 1e-2 of the output scale (3 bf16 GEMM/activation roundings deep)."""
    from oracle import fastvim_mixer_oracle
    c = load_golden("mixer.pt")[case]
    sd = _sd(c)
    m = _build(sd, c["token_size"])
    h = c["hidden"].cuda().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(h)
    assert y.dtype == torch.bfloat16
    hc = c["hidden"].clone().requires_grad_()
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    yref = fastvim_mixer_oracle(p, hc, c["token_size"], compute_dtype=F64, out_dtype=F64)
    s = max(1.0, yref.abs().max().item())
    assert _err(y, yref) <= 2e-2 * s, _err(y, yref)
    g = c["g"]
    y.backward(g.cuda().bfloat16())
    yref.backward(g.double())
    assert _err(h.grad, hc.grad) <= 3e-2 * max(1.0, hc.grad.abs().max().item())
    params = dict(m.named_parameters())
    for k in ("in_proj.weight", "out_proj.weight", "A_log", "D_b", "x_proj.weight", "conv1d.weight",
              "dt_proj_b.bias", "layernorm.weight", "conv1d_b.bias", "dt_proj.weight"):
        gr = p[k].grad
        e = _err(params[k].grad, gr)
        assert e <= 4e-2 * max(1.0, gr.abs().max().item()), (k, e, gr.abs().max().item())


def test_mixer_options_vs_oracle():
    """no norm after SSM, scaling_factor, init_layer_scale, conv without bias, in/out proj bias."""
    from fastvim_amd.mamba_simple_faster import Mamba
    from oracle import fastvim_mixer_oracle
    torch.manual_seed(0)
    for kw in (dict(use_norm_after_ssm=False), dict(scaling_factor=0.25), dict(init_layer_scale=0.1),
               dict(conv_bias=False), dict(bias=True)):
        m = Mamba(64, token_size=[5, 6], **kw).cuda()
        with torch.no_grad():
            for n, p_ in m.named_parameters():
                if n in ("D", "D_b") or n.endswith("bias"):
                    p_.add_(0.1 * torch.randn_like(p_))
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        h = torch.randn(2, 30, 64)
        hg = h.cuda().requires_grad_()
        y = m(hg)
        p = {k: v.clone().requires_grad_() for k, v in sd.items()}
        hc = h.clone().requires_grad_()
        okw = dict(use_norm_after_ssm=kw.get("use_norm_after_ssm", True), scaling_factor=kw.get("scaling_factor", 1))
        yref = fastvim_mixer_oracle(p, hc, (5, 6), compute_dtype=F64, out_dtype=F64, **okw)
        assert _err(y, yref) <= 1e-5 * max(1.0, yref.abs().max().item()), (kw, _err(y, yref))
        g = torch.randn_like(h)
        y.backward(g.cuda()); yref.backward(g.double())
        assert _err(hg.grad, hc.grad) <= 2e-5 * max(1.0, hc.grad.abs().max().item()), kw
        for n, q in m.named_parameters():
            e = _err(q.grad, p[n].grad)
            assert e <= 1e-4 * max(1.0, p[n].grad.abs().max().item()), (kw, n, e)


@pytest.mark.parametrize("d_model,grid,transposed", [(64, (4, 7), False), (192, (14, 14), False), (192, (14, 14), True),
                                                     (96, (3, 5), True), (768, (2, 16), False)])
def test_mixer_max_pool_vs_oracle(d_model, grid, transposed):
    """collapse_method="max" (mamba_simple_faster.py:299-305): forward and every gradient -- the pooled
    gradient is routed to the argmax column the forward kernel saved."""
    from fastvim_amd.mamba_simple_faster import Mamba
    from oracle import fastvim_mixer_oracle
    torch.manual_seed(d_model)
    rows, cols = grid
    m = Mamba(d_model, token_size=list(grid), collapse_method="max").cuda()
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n in ("D", "D_b", "layernorm.weight") or n.endswith("bias"):
                p_.add_(0.1 * torch.randn_like(p_))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    Bsz, Ltok = 2, rows * cols
    h = torch.randn(Bsz, Ltok, d_model)
    perm = (lambda t: t.reshape(Bsz, rows, cols, -1).transpose(1, 2).reshape(Bsz, Ltok, -1)) if transposed else (lambda t: t)
    hg = perm(h).contiguous().cuda().requires_grad_()
    y = m(hg, transposed_grid=transposed)
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    hc = h.clone().requires_grad_()
    yref = fastvim_mixer_oracle(p, hc, grid, collapse_method="max", compute_dtype=F64, out_dtype=F64)
    assert _err(y, perm(yref)) <= 1e-5 * max(1.0, yref.abs().max().item())
    g = torch.randn(Bsz, Ltok, d_model)
    y.backward(perm(g).contiguous().cuda())
    yref.backward(g.double())
    assert _err(hg.grad, perm(hc.grad)) <= 5e-5 * max(1.0, hc.grad.abs().max().item())
    for n, q in m.named_parameters():
        e = _err(q.grad, p[n].grad)
        assert e <= 2e-4 * max(1.0, p[n].grad.abs().max().item()), (n, e, p[n].grad.abs().max().item())


def test_mixer_full_size_deterministic_and_linear_in_out_proj():
    """BASELINE config 2 mixer shape (bs 128, 14x14, d 192): bitwise-reproducible fwd+bwd, and the
    block is linear in out_proj.weight (size-independent property)."""
    from fastvim_amd.mamba_simple_faster import Mamba
    torch.manual_seed(0)
    m = Mamba(192, token_size=[14, 14]).cuda()
    h = torch.randn(128, 196, 192, device="cuda", requires_grad=True)
    outs = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        h.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(h)
        y.float().square().mean().backward()
        outs.append((y.detach().clone(), h.grad.clone(), m.A_log.grad.clone(), m.conv1d.weight.grad.clone(),
                     m.x_proj_b.weight.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert all(torch.isfinite(t).all() for t in outs[0])
    with torch.no_grad():
        y1 = m(h)
        m.out_proj.weight.mul_(2.0)
        y2 = m(h)
    assert _err(y2, 2 * y1) <= 1e-5 * max(1.0, y2.abs().max().item())


@pytest.mark.parametrize("d_model,grid", [(384, (6, 7)), (768, (4, 14)), (96, (5, 3)), (512, (3, 16)),
                                          # long rows (cols a multiple of 8: the 512 / 1024 / 2048 px grids) take the
                                          # cell-walking conv kernels; d_inner = 1536 splits its channels over two blocks
                                          (192, (3, 32)), (768, (2, 24)), (64, (2, 40))])
def test_mixer_wide_models_vs_oracle(d_model, grid):
    """FastVim-S / -B widths (2 and 4 waves per pooling row, cross-wave LayerNorm statistics) and
    widths that take the generic lane mapping, fp32, forward + all gradients against the fp64 oracle."""
    from fastvim_amd.mamba_simple_faster import Mamba
    from oracle import fastvim_mixer_oracle
    torch.manual_seed(d_model)
    m = Mamba(d_model, token_size=list(grid)).cuda()
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n in ("D", "D_b", "layernorm.weight") or n.endswith("bias"):
                p_.add_(0.1 * torch.randn_like(p_))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    Ltok = grid[0] * grid[1]
    h = torch.randn(2, Ltok, d_model)
    for transposed in (False, True):
        m.zero_grad(set_to_none=True)
        rows, cols = grid
        perm = (lambda t: t.reshape(2, rows, cols, -1).transpose(1, 2).reshape(2, Ltok, -1)) if transposed else (lambda t: t)
        hg = perm(h).contiguous().cuda().requires_grad_()
        y = m(hg, transposed_grid=transposed)
        p = {k: v.clone().requires_grad_() for k, v in sd.items()}
        hc = h.clone().requires_grad_()
        yref = fastvim_mixer_oracle(p, hc, grid, compute_dtype=F64, out_dtype=F64)
        assert _err(y, perm(yref)) <= 2e-5 * max(1.0, yref.abs().max().item()), _err(y, perm(yref))
        g = torch.randn(2, Ltok, d_model)
        y.backward(perm(g).contiguous().cuda())
        yref.backward(g.double())
        assert _err(hg.grad, perm(hc.grad)) <= 5e-5 * max(1.0, hc.grad.abs().max().item())
        for n, q in m.named_parameters():
            e = _err(q.grad, p[n].grad)
            assert e <= 2e-4 * max(1.0, p[n].grad.abs().max().item()), (d_model, transposed, n, e, p[n].grad.abs().max().item())


def test_mixer_denormal_range_activations_vs_oracle():
    """Mixer-level twin of tests/test_scan_gpu.py::test_scan_denormal_range_inputs_vs_oracle: the fused row / scan kernels
    are built with fp32 denormals flushed (fastvim_amd/build.py).  A d_model = 192 mixer whose input has half of its images
    scaled into the 1e-18 ... 1e-22 range -- the pooled conv output, delta * B * u, the scan states and their adjoints are
    then fp32 denormals for those images -- against the fp64 oracle, forward and input gradient, at the usual fp32 bounds
    on the output scale; the O(1) images of the same batch must be unaffected."""
    from fastvim_amd.mamba_simple_faster import Mamba
    from oracle import fastvim_mixer_oracle
    torch.manual_seed(3)
    m = Mamba(192, token_size=[14, 14]).cuda()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    h = torch.randn(4, 196, 192)
    h[1] *= 1e-18
    h[3] *= 1e-22
    hg = h.cuda().requires_grad_()
    y = m(hg)
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    hc = h.clone().requires_grad_()
    yref = fastvim_mixer_oracle(p, hc, (14, 14), compute_dtype=F64, out_dtype=F64)
    assert torch.isfinite(y).all()
    assert _err(y, yref) <= 1e-5 * max(1.0, yref.abs().max().item()), _err(y, yref)
    for b in (0, 2):        # the ordinary images on their own scale
        assert _err(y[b], yref[b]) <= 1e-5 * max(1.0, yref[b].abs().max().item())
    g = torch.randn_like(h)
    y.backward(g.cuda()); yref.backward(g.double())
    assert torch.isfinite(hg.grad).all()
    assert _err(hg.grad, hc.grad) <= 2e-5 * max(1.0, hc.grad.abs().max().item())
    for n, q in m.named_parameters():
        e = _err(q.grad, p[n].grad)
        assert e <= 1e-4 * max(1.0, p[n].grad.abs().max().item()), (n, e)
