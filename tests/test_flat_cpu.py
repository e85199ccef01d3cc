"""Host logic of the flat training state (fastvim_amd/flat.py) that needs no GPU: bf16 shadow freshness,
checkpoint round trip of the fused optimizer's state, queue hygiene, restored process-wide switches."""
import warnings

import pytest
import torch


def _tiny():
    from fastvim_amd.fastvim import VisionMamba
    torch.manual_seed(0)
    return VisionMamba(img_size=32, patch_size=16, depth=2, embed_dim=32, channels=3, num_classes=5, rms_norm=True,
                       residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True,
                       drop_path_rate=0.0)


def test_shadow_follows_load_state_dict_and_in_place_writes():
    from fastvim_amd.flat import FlatTrainingState
    from fastvim_amd.mamba_simple_faster import _shadow
    m = _tiny()
    with FlatTrainingState(m) as flat:
        assert torch.equal(flat.shadow_flat.float(), flat.param_flat.bfloat16().float())
        sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
        m.load_state_dict(sd)                                   # post-hook re-casts the whole shadow
        assert torch.equal(flat.shadow_flat.float(), flat.param_flat.bfloat16().float())
        w = m.layers[0].mixer.in_proj.weight
        assert torch.equal(_shadow(w, torch.bfloat16).float(), sd["layers.0.mixer.in_proj.weight"].bfloat16().float())
        with torch.no_grad():                                   # a torch optimizer / EMA copy-in: version counter
            w.mul_(3.0)
        assert torch.equal(_shadow(w, torch.bfloat16).float(), w.detach().bfloat16().float())
        assert _shadow(w, torch.bfloat16).data_ptr() == w._fv_shadow.data_ptr()


def test_optimizer_state_dict_round_trip_and_ema_keys(tmp_path):
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState, load_checkpoint, save_checkpoint
    m1, m2 = _tiny(), _tiny()
    with FlatTrainingState(m1) as f1, FlatTrainingState(m2) as f2:
        o1 = FlatAdamW(f1, m1, lr=3e-4, betas=(0.9, 0.95), weight_decay=0.1, ema_decay=0.99)
        o2 = FlatAdamW(f2, m2, lr=1.0, ema_decay=0.99)
        with torch.no_grad():
            o1.exp_avg.normal_(); o1.exp_avg_sq.uniform_(); o1.ema.normal_(); o1.step_t.fill_(17.0)
            f1.param_flat.normal_()
        f1.refresh_shadow()
        path = str(tmp_path / "last.ckpt")
        ck = save_checkpoint(path, m1, o1, epoch=3)           # reference Lightning layout: state_dict / state_dict_ema
        load_checkpoint(path, m2, o2)
        named1, named2 = dict(m1.named_parameters()), dict(m2.named_parameters())
        for n in f1.names:                                      # compared by NAME: no dependence on flat offsets
            o = f1.offsets[n]
            k = named1[n].numel()
            assert torch.equal(named1[n], named2[n])
            for a, b in ((o1.exp_avg, o2.exp_avg), (o1.exp_avg_sq, o2.exp_avg_sq), (o1.ema, o2.ema)):
                assert torch.equal(a[o:o + k], b[f2.offsets[n]:f2.offsets[n] + k])
        assert o2.step_t.item() == 17.0 and abs(o2.lr.item() - 3e-4) < 1e-10 and o2.betas == (0.9, 0.95)
        assert torch.equal(f2.shadow_flat.float(), f2.param_flat.bfloat16().float())
        # reference layout (supervised_imagenet.py:107-114): `state_dict` behind "backbone.", `state_dict_ema` =
        # ModelEmaV2.module.state_dict(), i.e. UNPREFIXED -- its on_load_checkpoint feeds it to load_state_dict as is
        assert set(ck["state_dict"]) == {"backbone." + k for k in m1.state_dict()}
        assert set(ck["state_dict_ema"]) == set(m1.state_dict())
        _tiny().load_state_dict(ck["state_dict_ema"], strict=True)
        # the EMA weights load into a model (what MM_FastVim.load_pretrained prefers)
        load_checkpoint(ck, m2, use_ema=True)
        assert torch.equal(named2["head.weight"], o1.ema_state_dict()["head.weight"])
        # a dict shaped like the reference's own checkpoint, and one whose EMA entry carries the prefix as well
        ref_like = {"state_dict": {"backbone." + k: v.clone() for k, v in m1.state_dict().items()},
                    "state_dict_ema": {k: v.clone() + 1.0 for k, v in m1.state_dict().items()}}
        load_checkpoint(ref_like, m2, use_ema=True)
        assert torch.equal(named2["head.weight"], named1["head.weight"] + 1.0)
        both = dict(ref_like, state_dict_ema={"backbone." + k: v + 1.0 for k, v in ref_like["state_dict_ema"].items()})
        load_checkpoint(both, m2, use_ema=True)
        assert torch.equal(named2["head.weight"], both["state_dict_ema"]["backbone.head.weight"])


def test_zero_grad_warns_about_unfinished_backward_and_close_restores_switches():
    from fastvim_amd.flat import FlatTrainingState
    from fastvim_amd.mamba_simple_faster import _GroupedWgrad, _SideStream
    from fastvim_amd.mixer_ops import _Deferred
    before = (_Deferred.enabled, _GroupedWgrad.enabled, _SideStream.enabled)
    m = _tiny()
    flat = FlatTrainingState(m)
    assert _Deferred.enabled and _GroupedWgrad.enabled
    _GroupedWgrad.jobs.append(("g", "a", flat.grad_flat[:8], 1))          # a backward pass that was never finished
    with pytest.warns(RuntimeWarning, match="unfinished backward"):
        flat.zero_grad()
    assert not _GroupedWgrad.jobs
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        flat.zero_grad()                                                   # clean: no warning
    flat.close()
    assert (_Deferred.enabled, _GroupedWgrad.enabled, _SideStream.enabled) == before
    flat.close()                                                           # idempotent
