"""fv_mixer_xproj_bwd2 -- the x_proj adjoint (selective_scan_interface.py:698-734) -- against a plain fp32 torch reference of
the same op: the lane-per-channel kernel (weight-gradient partials requested) and the fp32-matrix-core data half (round 6,
taken when the weight gradient is left to the grouped launch), both directions, chunk partials summed inside."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,d_in,W,nchunks", [(1792, 1536, 80, 1), (1792, 1536, 80, 8), (448, 768, 56, 4), (7168, 768, 56, 1),
                                               (1000, 384, 44, 2), (300, 768, 44, 2), (70, 64, 36, 1), (130, 96, 34, 3), (130, 128, 34, 3), (61, 192, 38, 2), (64, 2048, 96, 1),
                                               (200, 1280, 112, 1)])
@pytest.mark.parametrize("dw", [False, True])
def test_xproj_bwd_vs_torch_fp32(M, d_in, W, nchunks, dw):
    from fastvim_amd import mixer_ops as Mo
    g = torch.Generator().manual_seed(M + W + nchunks)
    chunks = torch.randn(nchunks, 2, M, W, generator=g).cuda()
    xc = torch.randn(2, M, d_in, generator=g).cuda().to(torch.bfloat16)
    Wx = [(torch.randn(W, d_in, generator=g) * d_in ** -0.5).cuda() for _ in range(2)]
    dxc0 = torch.randn(2, M, d_in, generator=g).cuda()
    dxc = dxc0.clone()
    old = Mo._XPROJ_PRESUM
    Mo._XPROJ_PRESUM = 1 << 30          # the kernel sums the chunk partials itself
    try:
        out = Mo.xproj_bwd(chunks, xc.view(2, 1, M, d_in), Wx[0], Wx[1], dxc.view(2, 1, M, d_in), dw=dw)
    finally:
        Mo._XPROJ_PRESUM = old
    torch.cuda.synchronize()
    G = chunks.double().sum(0)                                     # (2, M, W)
    ref = dxc0.double() + torch.stack([G[k] @ Wx[k].double() for k in range(2)])
    err = (dxc.double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), err
    if dw:
        dW_ref = torch.stack([G[k].t() @ xc[k].double() for k in range(2)])
        assert (out.double() - dW_ref).abs().max().item() <= 3e-5 * dW_ref.abs().max().item()
    else:
        WP = (W + 7) // 8 * 8
        assert out.shape == (2, M, WP) and out.dtype == torch.bfloat16
        assert torch.equal(out[:, :, :W], chunks.sum(0).to(torch.bfloat16)) or \
            (out[:, :, :W].float() - G.float()).abs().max().item() <= 2 ** -8 * G.abs().max().item()
        assert (out[:, :, W:] == 0).all()
    # run to run: bitwise
    dxc2 = dxc0.clone()
    Mo._XPROJ_PRESUM = 1 << 30
    try:
        Mo.xproj_bwd(chunks, xc.view(2, 1, M, d_in), Wx[0], Wx[1], dxc2.view(2, 1, M, d_in), dw=dw)
    finally:
        Mo._XPROJ_PRESUM = old
    assert torch.equal(dxc2, dxc)


@pytest.mark.parametrize("M,d_in,W,nchunks", [(1792, 1536, 80, 1), (1792, 1536, 80, 8), (448, 768, 56, 4), (7168, 768, 56, 1),
                                               (64, 2048, 96, 1), (200, 1280, 112, 1), (130, 1024, 64, 3), (61, 768, 80, 2)])
def test_xproj_bwd_bf16_matrix_cores_vs_fp64_of_the_rounded_operands(M, d_in, W, nchunks):
    """fv_mixer_xproj_bwd3 (round 6): bf16(summed d x_dbl) @ bf16(weight), fp32 accumulate -- against fp64 of the SAME rounded
    operands (the only error left is the fp32 accumulation) and, loosely, against the exact-operand product the fp32 kernels
    compute (bf16 rounding of both operands); the published bf16 rows are the fp32 form's, bit for bit; repeat bitwise."""
    from fastvim_amd import mixer_ops as Mo
    g = torch.Generator().manual_seed(M + W + nchunks)
    chunks = torch.randn(nchunks, 2, M, W, generator=g).cuda()
    xc = torch.randn(2, M, d_in, generator=g).cuda().to(torch.bfloat16)
    Wx = [(torch.randn(W, d_in, generator=g) * d_in ** -0.5).cuda() for _ in range(2)]
    Wt = torch.stack([w.to(torch.bfloat16).t().contiguous() for w in Wx])          # (2, d_in, W) bf16
    dxc0 = torch.randn(2, M, d_in, generator=g).cuda()
    old = Mo._XPROJ_PRESUM
    Mo._XPROJ_PRESUM = 1 << 30
    try:
        dxc = dxc0.clone()
        out = Mo.xproj_bwd(chunks, xc.view(2, 1, M, d_in), Wx[0], Wx[1], dxc.view(2, 1, M, d_in), dw=False, Wx2_t=Wt)
        dxc_f = dxc0.clone()
        out_f = Mo.xproj_bwd(chunks, xc.view(2, 1, M, d_in), Wx[0], Wx[1], dxc_f.view(2, 1, M, d_in), dw=False)
        dxc2 = dxc0.clone()
        Mo.xproj_bwd(chunks, xc.view(2, 1, M, d_in), Wx[0], Wx[1], dxc2.view(2, 1, M, d_in), dw=False, Wx2_t=Wt)
    finally:
        Mo._XPROJ_PRESUM = old
    torch.cuda.synchronize()
    assert torch.equal(out, out_f) and torch.equal(dxc2, dxc)
    Gb = out[:, :, :W].double()                                                     # the rounded rows
    prod = torch.stack([Gb[k] @ Wt[k].double().t() for k in range(2)])
    ref = dxc0.double() + prod
    assert (dxc.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert (dxc - dxc_f).abs().max().item() <= 2e-2 * prod.abs().max().item()
    # the product as its own bf16 tensor: d xc untouched, dxc2 = bf16(product) (the rounding of the same fp32 accumulators)
    Mo._XPROJ_PRESUM = 1 << 30
    try:
        dxc3 = dxc0.clone()
        d2 = torch.full((2, M, d_in), float("nan"), device="cuda", dtype=torch.bfloat16)
        out3 = Mo.xproj_bwd(chunks, xc.view(2, 1, M, d_in), Wx[0], Wx[1], dxc3.view(2, 1, M, d_in), dw=False, Wx2_t=Wt, dxc2=d2)
    finally:
        Mo._XPROJ_PRESUM = old
    assert torch.equal(dxc3, dxc0) and torch.equal(out3, out)
    assert torch.equal(d2, (dxc - dxc0).to(torch.bfloat16)) or \
        (d2.double() - prod).abs().max().item() <= 2 ** -8 * prod.abs().max().item()


def test_wide_model_flat_step_with_and_without_the_second_addend():
    """A flat-state FastVim-B-width (d_model 384, d_inner 768, 2 blocks) training step: the x_proj adjoint's product handed to
    the conv + pool adjoint as a second bf16 addend (bf16 matrix cores, transposed shadow weight) against the fp32
    read-modify-write of d xc -- logits identical, gradients within bf16 rounding of the one product; the transposed shadow
    follows an optimizer step."""
    import copy
    from fastvim_amd import mamba_simple_faster as msf
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    torch.manual_seed(0)
    base = VisionMamba(img_size=224, depth=2, embed_dim=384, num_classes=20, rms_norm=True, residual_in_fp32=True,
                       fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.0).cuda().train()
    x = torch.randn(64, 3, 224, 224, device="cuda")
    gsel = torch.randn(64, 20, device="cuda")
    res = []
    for on in (True, False):
        m = copy.deepcopy(base)
        old = msf.XPROJ_TWO_ADDENDS
        msf.XPROJ_TWO_ADDENDS = on
        try:
            with FlatTrainingState(m) as flat:
                fv = m.layers[0].mixer.__dict__.get("_fv", {})
                assert ("Wx2_shadow_t" in fv) == True
                flat.zero_grad()
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = m(x)
                (y.float() * gsel).sum().backward()
                flat.finish_backward()
                torch.cuda.synchronize()
                res.append((y.detach().clone(), flat.grad_flat.clone()))
                if on:
                    opt = FlatAdamW(flat, m, lr=1e-2, weight_decay=0.05)
                    opt.step()
                    torch.cuda.synchronize()
                    ws, wt = fv["Wx2_shadow"], fv["Wx2_shadow_t"]
                    assert torch.equal(wt, ws.transpose(1, 2))
                    assert torch.equal(ws[0], m.layers[0].mixer.x_proj.weight.detach().to(torch.bfloat16))
        finally:
            msf.XPROJ_TWO_ADDENDS = old
    assert torch.equal(res[0][0], res[1][0])
    a, b = res[0][1], res[1][1]
    assert torch.isfinite(a).all()
    assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
    assert (a - b).norm().item() <= 5e-3 * b.norm().item()
