"""GPU parity of the two un-pooled baselines of the other task families (round 5, verdict item 8) against golden vectors
captured from the imported reference (tests/golden/gen_golden.py ``gen_baselines``): the Vim-encoder masked autoencoder
(models/mae/fastvim_mae.py) and ChannelVim with a middle class token
(models/channel_wise_tokenization/models_channel_mamba.py), both on the un-pooled Vim mixer's HIP kernels."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _err(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


@pytest.mark.parametrize("case", ["mae_vim_64_keep4", "mae_vim_96_keep9"])
def test_mae_vim_fp32_vs_reference_golden(case):
    from fastvim_amd.fastvim_mae import MaskedAutoencoderViM
    c = load_golden("baselines.pt")[case]
    cfg = c["cfg"]
    m = MaskedAutoencoderViM(img_size=cfg["img_size"], patch_size=cfg["patch_size"], depth=cfg["depth"],
                             embed_dim=cfg["embed_dim"], decoder_embed_dim=cfg["decoder_embed_dim"],
                             decoder_depth=cfg["decoder_depth"], rms_norm=True, residual_in_fp32=True,
                             fused_add_norm=True).cuda()
    m.load_state_dict(c["state_dict"], strict=True)
    loss, pred, mask = m(c["x"].cuda(), mask_ratio=0.75, noise=c["noise"].cuda())
    assert torch.equal(mask.cpu().float(), c["mask"].float())
    assert abs(loss.item() - c["loss"].item()) <= 2e-5 * max(1.0, abs(c["loss"].item()))
    assert _err(pred, c["pred"]) <= 2e-5 * max(1.0, c["pred"].abs().max().item()), _err(pred, c["pred"])
    loss.backward()
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 2e-4 * max(1e-3, gref.abs().max().item()), (k, e, gref.abs().max().item())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lb, pb, _ = m(c["x"].cuda(), mask_ratio=0.75, noise=c["noise"].cuda())
    assert abs(lb.item() - c["loss"].item()) <= 5e-2 * max(1.0, abs(c["loss"].item()))


@pytest.mark.parametrize("case", ["channelvim_64_c3", "channelvim_32_c5_spatial"])
def test_channelvim_fp32_vs_reference_golden(case):
    from fastvim_amd.models_channel_mamba import VisionMamba
    c = load_golden("baselines.pt")[case]
    m = VisionMamba(img_size=c["img"], patch_size=16, depth=4, embed_dim=32, channels=c["channels"], num_classes=10,
                    rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                    if_abs_pos_embed=True, if_cls_token=True, drop_path_rate=0.0, scan_order=c["scan_order"]).cuda().eval()
    m.load_state_dict(c["state_dict"], strict=True)
    logits = m(c["x"].cuda())
    assert _err(logits, c["logits"]) <= 2e-5 * max(1.0, c["logits"].abs().max().item()), _err(logits, c["logits"])
    logits.backward(c["g"].cuda())
    params = dict(m.named_parameters())
    for k, gref in c["grads"].items():
        e = _err(params[k].grad, gref)
        assert e <= 2e-4 * max(1e-3, gref.abs().max().item()), (k, e, gref.abs().max().item())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lb = m(c["x"].cuda())
    assert _err(lb, c["logits"]) <= 5e-2 * max(1.0, c["logits"].abs().max().item())


def test_channelvim_s_factory_runs_a_training_step_at_the_jumpcp_shape():
    """ChannelVim-S/16 (cell_imaging/config/ChannelVimS.yaml:24) at 8 channels, 224 px: 1 569 tokens through 24 un-pooled Vim
    blocks, bf16 autocast, HCS drawing a channel subset -- finite loss and gradients."""
    import random
    from fastvim_amd.models_channel_mamba import channelvim_small_patch16_224_final_pool_mean_abs_pos_embed_with_midclstok_div2 as f
    torch.manual_seed(0)
    random.seed(0)
    m = f(channels=8, num_classes=161, drop_path_rate=0.1).cuda().train()
    x = torch.randn(4, 8, 224, 224, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = m(x)
    assert y.shape == (4, 161)
    y.float().square().mean().backward()
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in m.parameters())
    assert m.cls_token.grad is not None and m.cls_token.grad.abs().max() > 0
