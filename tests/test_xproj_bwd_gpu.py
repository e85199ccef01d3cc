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
