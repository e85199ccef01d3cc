"""GPU parity of the MAE masked FastVim mixer: ``Mamba_masked`` and the two row kernels
behind it, against golden vectors captured from the imported reference
(mamba_ssm/modules/mamba_simple_masked_faster.py) and against the fp64 oracle."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.mark.parametrize("in_dt,out_dt", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16),
                                          (torch.bfloat16, torch.float32)])
def test_rows_segment_sum_and_gather_vs_torch(in_dt, out_dt):
    from fastvim_amd import mixer_ops as M
    torch.manual_seed(0)
    B, Lk, rows, d = 3, 11, 5, 192
    x = torch.randn(2, B, Lk, d, device="cuda").to(in_dt)
    idx = torch.randint(0, rows, (2, B, Lk), device="cuda", dtype=torch.int32)
    idx[0, 0, :] = 2                                   # a row that takes everything, rows that take nothing
    out = M.rows_segment_sum(x, idx, rows, 0.25, out_dtype=out_dt)
    ref = torch.zeros(2, B, rows, d, device="cuda", dtype=F64)
    ref.scatter_add_(2, idx.long()[..., None].expand(2, B, Lk, d), x.double())
    tol = 1e-6 if out_dt == torch.float32 and in_dt == torch.float32 else 2e-2
    assert _err(out, 0.25 * ref) <= tol * max(1.0, ref.abs().max().item())
    shared = M.rows_segment_sum(x[0].contiguous(), idx, rows, 1.0, out_dtype=torch.float32)       # one input, two index sets
    ref1 = torch.zeros(2, B, rows, d, device="cuda", dtype=F64)
    ref1.scatter_add_(2, idx.long()[..., None].expand(2, B, Lk, d), x[0].double()[None].expand(2, B, Lk, d))
    assert _err(shared, ref1) <= (1e-5 if in_dt == torch.float32 else 2e-2) * max(1.0, ref1.abs().max().item())
    y = torch.randn(2, B, rows, d, device="cuda").to(in_dt)
    gat = M.rows_gather(y, idx, 0.5, out_dtype=out_dt)
    refg = 0.5 * torch.gather(y.double(), 2, idx.long()[..., None].expand(2, B, Lk, d))
    assert _err(gat, refg) <= (1e-6 if out_dt == torch.float32 and in_dt == torch.float32 else 2e-2) * 4
    # the two kernels are each other's adjoint: <segsum(x), y> == <x, gather(y)>
    xs, ys = x.float(), y.float()
    lhs = (M.rows_segment_sum(xs, idx, rows).double() * ys.double()).sum()
    rhs = (xs.double() * M.rows_gather(ys, idx).double()).sum()
    assert abs(lhs - rhs).item() <= 1e-4 * max(1.0, abs(lhs).item())


@pytest.mark.parametrize("case", ["d32_4x4_keep6", "d32_3x5_keep9", "d32_4x4_keep7_unsorted", "d64_6x6_keep9"])
def test_masked_mixer_fp32_vs_reference_golden(case):
    from fastvim_amd.mamba_simple_masked_faster import Mamba_masked
    c = load_golden("masked.pt")[case]
    sd = c["state_dict"]
    m = Mamba_masked(sd["in_proj.weight"].shape[1], token_size=list(c["token_size"])).cuda()
    m.load_state_dict(sd, strict=True)
    hg = c["hidden"].cuda().requires_grad_()
    y = m(hg, c["ids_keep"].cuda())
    ref = c["out"]
    assert _err(y, ref) <= 1e-5 * max(1.0, ref.abs().max().item()), _err(y, ref)
    y.backward(c["g"].cuda())
    assert _err(hg.grad, c["dhidden"]) <= 2e-5 * max(1.0, c["dhidden"].abs().max().item())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 1e-4 * max(1.0, gref.abs().max().item()), (k, e, gref.abs().max().item())


@pytest.mark.parametrize("d_model,grid,keep,dtype", [
    (192, (14, 14), 49, torch.float32),          # FastVim-T MAE pre-training geometry: 25 % of 196 tokens kept
    (192, (14, 14), 49, torch.bfloat16),
    (384, (14, 14), 49, torch.bfloat16),
    (96, (6, 10), 15, torch.float32),
])
def test_masked_mixer_vs_oracle(d_model, grid, keep, dtype):
    from fastvim_amd.mamba_simple_masked_faster import Mamba_masked
    from oracle import masked_mixer_oracle
    torch.manual_seed(keep + d_model)
    rows, cols = grid
    m = Mamba_masked(d_model, token_size=list(grid)).cuda()
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n in ("D", "D_b", "layernorm.weight") or n.endswith("bias"):
                p_.add_(0.1 * torch.randn_like(p_))
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    Bsz = 4
    ids = torch.stack([torch.randperm(rows * cols)[:keep].sort().values for _ in range(Bsz)])
    h = torch.randn(Bsz, keep, d_model)
    g = torch.randn(Bsz, keep, d_model)
    bf = dtype == torch.bfloat16
    if bf:
        h, g = h.bfloat16().float(), g.bfloat16().float()
    p = {k: v.clone().requires_grad_() for k, v in sd.items()}
    hc = h.clone().requires_grad_()
    yref = masked_mixer_oracle(p, hc, ids, grid, compute_dtype=F64, out_dtype=F64)
    yref.backward(g.double())
    hg = h.cuda().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
        y = m(hg, ids.cuda())
    tol_y, tol_dh, tol_w = (2e-2, 3e-2, 4e-2) if bf else (2e-5, 5e-5, 2e-4)
    assert _err(y, yref) <= tol_y * max(1.0, yref.abs().max().item()), _err(y, yref)
    y.backward(g.cuda().to(y.dtype))
    assert _err(hg.grad, hc.grad) <= tol_dh * max(1.0, hc.grad.abs().max().item())
    for n, q in m.named_parameters():
        e = _err(q.grad, p[n].grad)
        assert e <= tol_w * max(1.0, p[n].grad.abs().max().item()), (n, e, p[n].grad.abs().max().item())
    # deterministic: same inputs, same bits
    m.zero_grad(set_to_none=True)
    hg2 = h.cuda().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
        y2 = m(hg2, ids.cuda())
    y2.backward(g.cuda().to(y2.dtype))
    assert torch.equal(y, y2) and torch.equal(hg.grad, hg2.grad)
