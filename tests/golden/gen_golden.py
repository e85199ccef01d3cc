"""Generate golden vectors from the imported reference (build container only).

    python tests/golden/gen_golden.py

Imports insitro/FastVim from /root/reference on CPU through ``_ref_import`` and
freezes inputs + reference outputs as small ``.pt`` fixtures next to this file.
The fixtures are data (inputs / expected outputs); nothing of the reference's
source is stored.  ``tests/test_oracle_golden.py`` pins ``oracle/`` against them
and the ``-m gpu`` tests pin the HIP path against them.

Input distributions follow the reference's own tests:
mamba-1p1p1/tests/ops/test_selective_scan.py:61-122 and
fastvim_kernel/mamba-1p1p1/tests/test_compressed_scan.py:54-127.
"""
import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

from _ref_import import load_reference  # noqa: E402

ref = load_reference()
from oracle.model import make_state_dict  # noqa: E402  (seeded parameter recipe only)


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------
def scan_case(batch, dim, L, N, groups=1, has_D=True, has_z=True, softplus=True, seed=0):
    torch.manual_seed(seed)
    A = -0.5 * torch.rand(dim, N)
    bshape = (batch, N, L) if groups == 1 else (batch, groups, N, L)
    B = torch.randn(*bshape)
    C = torch.randn(*bshape)
    D = torch.randn(dim) if has_D else None
    z = torch.randn(batch, dim, L) if has_z else None
    delta_bias = 0.5 * torch.rand(dim)
    u = torch.randn(batch, dim, L)
    delta = 0.5 * torch.rand(batch, dim, L)
    inp = dict(u=u, delta=delta, A=A, B=B, C=C, D=D, z=z, delta_bias=delta_bias)
    leaves = {k: v.clone().requires_grad_() for k, v in inp.items() if v is not None}
    out, last = ref.ssi.selective_scan_ref(
        leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves.get("D"),
        z=leaves.get("z"), delta_bias=leaves["delta_bias"], delta_softplus=softplus,
        return_last_state=True)
    g = torch.randn_like(out)
    out.backward(g)
    grads = {k: v.grad.clone() for k, v in leaves.items()}
    # bf16 I/O case: the reference rounds once at the end (selective_scan_interface.py:205)
    bf = {k: (v.bfloat16() if k in ("u", "delta", "B", "C", "z") and v is not None else v)
          for k, v in inp.items()}
    out_bf16 = ref.ssi.selective_scan_ref(bf["u"], bf["delta"], bf["A"], bf["B"], bf["C"], bf["D"],
                                          z=bf["z"], delta_bias=bf["delta_bias"],
                                          delta_softplus=softplus)
    return dict(inputs=inp, softplus=softplus, out=out.detach(), last_state=last.detach(), g=g,
                grads=grads, out_bf16=out_bf16)


def gen_scan():
    cases = {}
    for L in (14, 112, 128, 256):
        cases[f"b2_d4_L{L}_n8"] = scan_case(2, 4, L, 8)
    cases["b2_d8_L14_n16"] = scan_case(2, 8, 14, 16)
    cases["b2_d4_L128_n8_g2"] = scan_case(2, 4, 128, 8, groups=2)
    cases["b2_d4_L14_n8_noDz"] = scan_case(2, 4, 14, 8, has_D=False, has_z=False)
    cases["b2_d4_L64_n8_nosoftplus"] = scan_case(2, 4, 64, 8, softplus=False)
    cases["b1_d2_L2100_n8"] = scan_case(1, 2, 2100, 8)   # > one 2048 chunk of the reference kernel
    save("scan.pt", cases)


def gen_compressed_scan():
    # the fork's own ref == scan at Lc, repeat_interleave(cf), + D*u_full
    # (fastvim_kernel/.../faster_mamba_ssm/ops/selective_scan_interface.py:243-248); it cannot be
    # imported (its module imports the CUDA extension and prints), so the vectors are produced
    # with the main reference scan + exactly those two torch ops.
    cases = {}
    for cf, Lc, has_D in ((1, 6, True), (2, 64, True), (14, 14, True), (8, 32, False)):
        torch.manual_seed(0)
        batch, dim, N = 2, 4, 8
        L = Lc * cf
        A = -0.5 * torch.rand(dim, N)
        B = torch.randn(batch, N, Lc)
        C = torch.randn(batch, N, Lc)
        D = torch.randn(dim) if has_D else None
        delta_bias = 0.5 * torch.rand(dim)
        u_full = torch.randn(batch, dim, L)
        u_c = u_full.reshape(batch, dim, Lc, cf).mean(-1)
        delta = 0.5 * torch.rand(batch, dim, Lc)
        y, last = ref.ssi.selective_scan_ref(u_c, delta, A, B, C, None, None, delta_bias, False, True)
        out = y.repeat_interleave(cf, dim=2)
        if D is not None:
            out = out + u_full * D[:, None]
        cases[f"cf{cf}_Lc{Lc}_D{int(has_D)}"] = dict(
            inputs=dict(u_full=u_full, u_c=u_c, delta=delta, A=A, B=B, C=C, D=D, delta_bias=delta_bias),
            out=out, last_state=last)
    save("compressed_scan.pt", cases)


def gen_conv():
    cases = {}
    for name, (Bsz, D, L, W, act, has_bias) in {
        "b2_d8_L196_silu": (2, 8, 196, 4, "silu", True),
        "b2_d8_L5_silu": (2, 8, 5, 4, "silu", True),
        "b1_d4_L2_none": (1, 4, 2, 4, None, False),
        "b2_d6_L33_w3": (2, 6, 33, 3, "silu", True),
    }.items():
        torch.manual_seed(1)
        x = torch.randn(Bsz, D, L, requires_grad=True)
        w = torch.randn(D, W, requires_grad=True)
        b = torch.randn(D, requires_grad=True) if has_bias else None
        y = ref.causal_conv1d_fn(x, w, b, act)
        g = torch.randn_like(y)
        y.backward(g)
        cases[name] = dict(x=x.detach(), w=w.detach(), b=None if b is None else b.detach(), act=act,
                           y=y.detach(), g=g, dx=x.grad, dw=w.grad, db=None if b is None else b.grad)
    save("conv1d.pt", cases)


def gen_norm():
    cases = {}
    torch.manual_seed(2)
    for name, (rms, prenorm, has_res, xdt) in {
        "rms_prenorm_res": (True, True, True, torch.float32),
        "rms_prenorm_nores": (True, True, False, torch.float32),
        "rms_final_res": (True, False, True, torch.float32),
        "ln_prenorm_res": (False, True, True, torch.float32),
        "rms_prenorm_res_bf16": (True, True, True, torch.bfloat16),
    }.items():
        x = torch.randn(2, 20, 48).to(xdt)
        res = torch.randn(2, 20, 48) if has_res else None
        w = 1 + 0.1 * torch.randn(48)
        b = None if rms else 0.1 * torch.randn(48)
        fn = ref.rms_norm_fn if rms else ref.layer_norm_fn
        outs = fn(x, w, b, residual=res, prenorm=prenorm, residual_in_fp32=True, eps=1e-5)
        cases[name] = dict(x=x, residual=res, w=w, b=b, rms=rms, prenorm=prenorm, eps=1e-5, out=outs)
    save("norm.pt", cases)


def _mixer_case(d_model, token_size, Bsz, sd=None, seed=3):
    torch.manual_seed(seed)
    m = ref.msf.Mamba(d_model, token_size=list(token_size))
    if sd is not None:
        m.load_state_dict(sd, strict=True)
    else:
        # move the special inits off their symmetric defaults so every parameter matters
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n in ("D", "D_b", "layernorm.weight"):
                    p.add_(0.2 * torch.randn_like(p))
                elif n in ("layernorm.bias", "conv1d.bias", "conv1d_b.bias"):
                    p.add_(0.1 * torch.randn_like(p))
                elif n in ("A_log", "A_b_log"):
                    p.add_(0.1 * torch.randn_like(p))
    L = token_size[0] * token_size[1]
    h = torch.randn(Bsz, L, d_model, requires_grad=True)
    y = m(h)
    g = torch.randn_like(y)
    y.backward(g)
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    return m, dict(hidden=h.detach(), out=y.detach(), g=g, dhidden=h.grad.clone(), grads=grads,
                   token_size=tuple(token_size))


def gen_mixer():
    cases = {}
    m, c = _mixer_case(32, (4, 4), 2)
    c["state_dict"] = {k: v.clone() for k, v in m.state_dict().items()}
    cases["d32_4x4"] = c
    m, c = _mixer_case(32, (3, 5), 2, seed=4)         # non-square grid
    c["state_dict"] = {k: v.clone() for k, v in m.state_dict().items()}
    cases["d32_3x5"] = c
    # real FastVim-T mixer; parameters from the seeded recipe (not stored)
    sd = make_state_dict(seed=11, embed_dim=192, depth=1)
    msd = {k[len("layers.0.mixer."):]: v for k, v in sd.items() if k.startswith("layers.0.mixer.")}
    m, c = _mixer_case(192, (14, 14), 1, sd=msd, seed=5)
    keep = ("in_proj.weight", "A_log", "A_b_log", "D", "D_b", "x_proj.weight", "conv1d.weight",
            "conv1d_b.bias", "dt_proj_b.bias", "layernorm.weight")
    c["grads"] = {k: v for k, v in c["grads"].items() if k in keep and v.numel() <= 20000}
    c["param_recipe"] = dict(seed=11, embed_dim=192, depth=1, prefix="layers.0.mixer.")
    cases["d192_14x14"] = c
    save("mixer.pt", cases)


def gen_model():
    cases = {}
    fv = ref.fastvim
    for name, img in (("tiny_64x64", (64, 64)), ("tiny_48x80", (48, 80))):
        torch.manual_seed(6)
        model = fv.VisionMamba(img_size=img, patch_size=16, depth=4, embed_dim=32, channels=3,
                               num_classes=10, rms_norm=True, residual_in_fp32=True,
                               fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True,
                               drop_path_rate=0.0)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith(("D", "D_b", "norm.weight", "layernorm.weight", "norm_f.weight")):
                    p.add_(0.2 * torch.randn_like(p))
                elif n.endswith(("layernorm.bias", "head.bias", "patch_embed.proj.bias")):
                    p.add_(0.1 * torch.randn_like(p))
        model.eval()
        x = torch.randn(2, 3, *img)
        hiddens = []
        hooks = [l.register_forward_hook(lambda mod, i, o: hiddens.append(o[0].detach().clone()))
                 for l in model.layers]
        logits = model(x)
        for hk in hooks:
            hk.remove()
        g = torch.randn_like(logits)
        logits.backward(g)
        grads = {n: p.grad.clone() for n, p in model.named_parameters()
                 if n in ("pos_embed", "head.weight", "layers.0.mixer.in_proj.weight",
                          "layers.1.mixer.A_b_log", "layers.3.mixer.x_proj_b.weight",
                          "layers.2.norm.weight", "patch_embed.proj.bias", "norm_f.weight",
                          "layers.1.mixer.conv1d.weight", "layers.2.mixer.dt_proj.bias")}
        cases[name] = dict(img=img, state_dict={k: v.clone() for k, v in model.state_dict().items()},
                           x=x, logits=logits.detach(), hiddens=hiddens, g=g, grads=grads,
                           cfg=dict(patch_size=16, depth=4, embed_dim=32, num_classes=10))
    save("model_tiny.pt", cases)

    # FastVim-T, bs=2 (BASELINE config 1): parameters and pixels from seeded recipes
    sd = make_state_dict(seed=7, embed_dim=192, depth=24)
    model = fv.vim_tiny_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(drop_path_rate=0.0)
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(123))
    logits = model(x)
    g = torch.randn(logits.shape, generator=torch.Generator().manual_seed(124))
    logits.backward(g)
    grads = {n: p.grad.clone() for n, p in model.named_parameters()
             if n in ("head.bias", "norm_f.weight", "layers.0.norm.weight", "layers.23.mixer.D_b",
                      "layers.12.mixer.A_log", "layers.5.mixer.dt_proj.bias", "patch_embed.proj.bias",
                      "layers.7.mixer.conv1d_b.weight", "layers.0.mixer.x_proj.weight")}
    save("model_fastvim_t.pt", dict(param_seed=7, x_seed=123, g_seed=124, x_probe=x[0, 0, :2, :8].clone(),
                                     logits=logits.detach(), grads=grads))


def gen_channel():
    """FastChannelVim (BASELINE config 5 family): channel mixer + tiny channel backbone, Channel-First."""
    import random
    cases = {}
    # ---- mixer: mamba_simple_channel_faster.Mamba.forward(hidden, tokens_per_patch)
    for name, d_model, ts, tpp, Bsz, seed in (("d32_4x4_t3", 32, (4, 4), 3, 2, 21), ("d32_2x6_t5", 32, (2, 6), 5, 2, 22),
                                              ("d64_4x2_t8", 64, (4, 2), 8, 1, 23)):
        torch.manual_seed(seed)
        m = ref.mscf.Mamba(d_model, token_size=list(ts))
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n in ("D", "D_b", "layernorm.weight"):
                    p.add_(0.2 * torch.randn_like(p))
                elif n in ("layernorm.bias", "conv1d.bias", "conv1d_b.bias"):
                    p.add_(0.1 * torch.randn_like(p))
                elif n in ("A_log", "A_b_log"):
                    p.add_(0.1 * torch.randn_like(p))
        h = torch.randn(Bsz, ts[0] * ts[1] * tpp, d_model, requires_grad=True)
        y = m(h, tpp)
        g = torch.randn_like(y)
        y.backward(g)
        cases["mixer_" + name] = dict(
            hidden=h.detach(), out=y.detach(), g=g, dhidden=h.grad.clone(), token_size=ts, tokens_per_patch=tpp,
            grads={n: p.grad.clone() for n, p in m.named_parameters()},
            state_dict={k: v.clone() for k, v in m.state_dict().items()})
    # ---- backbone, eval mode (HCS off) and one training-mode HCS draw
    for name, img, chans, train in (("tiny_64x64_c3", (64, 64), 3, False), ("tiny_64x96_c5_hcs", (64, 96), 5, True)):
        torch.manual_seed(31)
        model = ref.chan.VisionMamba(img_size=img, patch_size=16, depth=4, embed_dim=32, channels=chans,
                                     num_classes=10, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                                     final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.0)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith(("D", "D_b", "norm.weight", "layernorm.weight", "norm_f.weight")):
                    p.add_(0.2 * torch.randn_like(p))
                elif n.endswith(("layernorm.bias", "head.bias", "patch_embed.proj.bias")):
                    p.add_(0.1 * torch.randn_like(p))
        model.train(train)
        x = torch.randn(2, chans, *img)
        random.seed(1234)
        # record the HCS subset the reference draws (PatchEmbedPerChannel.forward :167-185)
        st = random.getstate()
        if train:
            k = random.randint(1, chans)
            subset = sorted(random.sample(range(chans), k=k))
        else:
            subset = list(range(chans))
        random.setstate(st)
        logits = model(x)
        g = torch.randn_like(logits)
        logits.backward(g)
        grads = {n: p.grad.clone() for n, p in model.named_parameters()
                 if n in ("pos_embed", "head.weight", "layers.0.mixer.in_proj.weight", "layers.1.mixer.A_b_log",
                          "layers.3.mixer.x_proj_b.weight", "layers.2.norm.weight", "patch_embed.proj.bias",
                          "patch_embed.proj.weight", "patch_embed.channel_embed.weight", "norm_f.weight",
                          "layers.1.mixer.conv1d.weight", "layers.2.mixer.dt_proj.bias")}
        cases[name] = dict(img=img, channels=chans, train=train, py_seed=1234, subset=subset,
                           state_dict={k: v.clone() for k, v in model.state_dict().items()},
                           x=x, logits=logits.detach(), g=g, grads=grads,
                           cfg=dict(patch_size=16, depth=4, embed_dim=32, num_classes=10))
    save("channel.pt", cases)


def gen_vim():
    """Vim baseline (models/vim.py + mamba_simple.py, bidirectional v2 with LayerNorm after the SSM): mixer and a
    tiny backbone with the middle class token.  The reference's fused CUDA path is replaced by its own reference
    path (use_fast_path=False: causal_conv1d_fn + selective_scan_ref), which computes the same function."""
    cases = {}
    for name, d_model, L, Bsz, seed in (("mixer_d32_L9", 32, 9, 2, 41), ("mixer_d32_L13", 32, 13, 2, 42),
                                        ("mixer_d64_L20", 64, 20, 1, 43)):
        torch.manual_seed(seed)
        m = ref.ms.Mamba(d_model, use_fast_path=False)
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n in ("D", "D_b", "layernorm.weight"):
                    p.add_(0.2 * torch.randn_like(p))
                elif n in ("layernorm.bias", "conv1d.bias", "conv1d_b.bias"):
                    p.add_(0.1 * torch.randn_like(p))
                elif n in ("A_log", "A_b_log"):
                    p.add_(0.1 * torch.randn_like(p))
        h = torch.randn(Bsz, L, d_model, requires_grad=True)
        y = m(h)
        g = torch.randn_like(y)
        y.backward(g)
        cases[name] = dict(hidden=h.detach(), out=y.detach(), g=g, dhidden=h.grad.clone(),
                           grads={n: p.grad.clone() for n, p in m.named_parameters()},
                           state_dict={k: v.clone() for k, v in m.state_dict().items()})
    torch.manual_seed(51)
    model = ref.vim.VisionMamba(img_size=64, patch_size=16, depth=4, embed_dim=32, channels=3, num_classes=10,
                                rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                                if_abs_pos_embed=True, if_cls_token=True, use_middle_cls_token=True,
                                drop_path_rate=0.0, ssm_cfg={"use_fast_path": False})
    init_sd = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(("D", "D_b", "norm.weight", "layernorm.weight", "norm_f.weight")):
                p.add_(0.2 * torch.randn_like(p))
            elif n.endswith(("layernorm.bias", "head.bias", "patch_embed.proj.bias")):
                p.add_(0.1 * torch.randn_like(p))
    model.eval()
    x = torch.randn(2, 3, 64, 64)
    logits = model(x)
    g = torch.randn_like(logits)
    logits.backward(g)
    grads = {n: p.grad.clone() for n, p in model.named_parameters()
             if n in ("pos_embed", "cls_token", "head.weight", "layers.0.mixer.in_proj.weight", "layers.1.mixer.A_b_log",
                      "layers.3.mixer.x_proj_b.weight", "layers.2.norm.weight", "patch_embed.proj.bias", "norm_f.weight",
                      "layers.1.mixer.conv1d.weight", "layers.2.mixer.dt_proj.bias")}
    cases["tiny_64x64_cls"] = dict(img=(64, 64), init_seed=51, init_probe={k: init_sd[k] for k in
                                   ("cls_token", "layers.0.mixer.in_proj.weight", "layers.3.mixer.dt_proj.bias", "pos_embed")},
                                   state_dict={k: v.clone() for k, v in model.state_dict().items()},
                                   x=x, logits=logits.detach(), g=g, grads=grads,
                                   cfg=dict(patch_size=16, depth=4, embed_dim=32, num_classes=10))
    save("vim.pt", cases)


def _masked_case(d_model, token_size, Bsz, n_keep, seed, sorted_ids=True):
    """MAE masked mixer (mamba_simple_masked_faster.py:167-325): hidden holds only the kept tokens, ids_keep their
    positions in the full rows x cols grid (ascending per sample, as random_masking / the MAE Block produce them,
    models_mamba_faster_mae_vimdecoder.py:381-386, 757)."""
    torch.manual_seed(seed)
    m = ref.msmf.Mamba_masked(d_model, token_size=list(token_size))
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n in ("D", "D_b", "layernorm.weight"):
                p.add_(0.2 * torch.randn_like(p))
            elif n in ("layernorm.bias", "conv1d.bias", "conv1d_b.bias", "A_log", "A_b_log"):
                p.add_(0.1 * torch.randn_like(p))
    L = token_size[0] * token_size[1]
    ids = torch.stack([torch.randperm(L)[:n_keep] for _ in range(Bsz)])
    if sorted_ids:
        ids = ids.sort(dim=1).values
    h = torch.randn(Bsz, n_keep, d_model, requires_grad=True)
    y = m(h, ids.clone())
    g = torch.randn_like(y)
    y.backward(g)
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    return dict(hidden=h.detach(), ids_keep=ids, out=y.detach(), g=g, dhidden=h.grad.clone(), grads=grads,
                token_size=tuple(token_size), state_dict={k: v.clone() for k, v in m.state_dict().items()})


def gen_masked():
    cases = {
        "d32_4x4_keep6": _masked_case(32, (4, 4), 2, 6, seed=21),
        "d32_3x5_keep9": _masked_case(32, (3, 5), 3, 9, seed=22),
        "d32_4x4_keep7_unsorted": _masked_case(32, (4, 4), 2, 7, seed=23, sorted_ids=False),
        "d64_6x6_keep9": _masked_case(64, (6, 6), 2, 9, seed=24),      # 25 % kept, some rows empty
    }
    save("masked.pt", cases)


def gen_mae():
    """Tiny MaskedAutoencoderViM (the mae_FastVim_* recipe: rms_norm, fused_add_norm, residual_in_fp32) with the
    masking noise captured: loss, prediction, mask and a few gradients of one pre-training step."""
    cases = {}
    for name, img, seed in (("tiny_64_keep4", 64, 31), ("tiny_96_keep9", 96, 32)):
        torch.manual_seed(seed)
        m = ref.mae.MaskedAutoencoderViM(img_size=img, patch_size=16, depth=4, embed_dim=32, decoder_embed_dim=32,
                                         decoder_depth=2, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                                         ssm_cfg={"use_fast_path": False})
        with torch.no_grad():                      # move the symmetric inits so every parameter matters
            for n, p in m.named_parameters():
                if n.endswith((".D", ".D_b", "layernorm.weight", "norm.weight", "norm_f.weight", "decoder_norm.weight")):
                    p.add_(0.1 * torch.randn_like(p))
                elif n.endswith("bias") and p.requires_grad:
                    p.add_(0.05 * torch.randn_like(p))
        x = torch.randn(2, 3, img, img)
        L = (img // 16) ** 2
        noise = torch.rand(2, L)
        rand = torch.rand
        torch.rand = lambda *a, **k: noise.clone()     # random_masking draws torch.rand(N, L, device=...) (:750)
        try:
            loss, pred, mask = m(x, mask_ratio=0.75)
        finally:
            torch.rand = rand
        loss.backward()
        keep = ("patch_embed.proj.weight", "layers.0.mixer.in_proj.weight", "layers.1.mixer.A_b_log", "layers.3.mixer.D",
                "layers.2.mixer.conv1d_b.weight", "norm_f.weight", "decoder_embed.weight", "mask_token",
                "decoder_blocks.1.mixer.x_proj.weight", "decoder_pred.bias")
        grads = {n: p.grad.clone() for n, p in m.named_parameters() if n in keep}
        cases[name] = dict(x=x, noise=noise, loss=loss.detach(), pred=pred.detach(), mask=mask, grads=grads,
                           state_dict={k: v.clone() for k, v in m.state_dict().items()},
                           cfg=dict(img_size=img, patch_size=16, depth=4, embed_dim=32, decoder_embed_dim=32, decoder_depth=2))
    # initialisation contract: parameters of a freshly built model for a fixed torch seed
    torch.manual_seed(77)
    m = ref.mae.MaskedAutoencoderViM(img_size=64, patch_size=16, depth=2, embed_dim=32, decoder_embed_dim=32,
                                     decoder_depth=1, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                                     ssm_cfg={"use_fast_path": False})
    cases["init_seed77"] = dict(state_dict={k: v.clone() for k, v in m.state_dict().items()},
                                cfg=dict(img_size=64, patch_size=16, depth=2, embed_dim=32, decoder_embed_dim=32, decoder_depth=1))
    save("mae.pt", cases)


def gen_baselines():
    """The two un-pooled baselines of the other task families (round 5, verdict item 8): the Vim-encoder MAE
    (models/mae/fastvim_mae.py: class token in the middle of the kept tokens, appended in the decoder) and ChannelVim
    (models/channel_wise_tokenization/models_channel_mamba.py: Channel-First tokens, middle class token).  Tiny models,
    reference path use_fast_path=False (causal_conv1d_fn + selective_scan_ref)."""
    cases = {}
    for name, img, seed in (("mae_vim_64_keep4", 64, 61), ("mae_vim_96_keep9", 96, 62)):
        torch.manual_seed(seed)
        m = ref.mae_vim.MaskedAutoencoderViM(img_size=img, patch_size=16, depth=4, embed_dim=32, decoder_embed_dim=32,
                                             decoder_depth=2, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                                             ssm_cfg={"use_fast_path": False})
        with torch.no_grad():
            for n, p in m.named_parameters():
                if n.endswith((".D", ".D_b", "layernorm.weight", "norm.weight", "norm_f.weight", "decoder_norm.weight")):
                    p.add_(0.1 * torch.randn_like(p))
                elif n.endswith("bias") and p.requires_grad:
                    p.add_(0.05 * torch.randn_like(p))
        x = torch.randn(2, 3, img, img)
        L = (img // 16) ** 2
        noise = torch.rand(2, L)
        rand = torch.rand
        torch.rand = lambda *a, **k: noise.clone()     # random_masking draws torch.rand(N, L, device=...) (fastvim_mae.py:551)
        try:
            loss, pred, mask = m(x, mask_ratio=0.75)
        finally:
            torch.rand = rand
        loss.backward()
        keep = ("patch_embed.proj.weight", "cls_token", "layers.0.mixer.in_proj.weight", "layers.1.mixer.A_b_log",
                "layers.3.mixer.D", "layers.2.mixer.conv1d_b.weight", "norm_f.weight", "decoder_embed.weight", "mask_token",
                "decoder_blocks.1.mixer.x_proj.weight", "decoder_pred.bias")
        cases[name] = dict(x=x, noise=noise, loss=loss.detach(), pred=pred.detach(), mask=mask,
                           grads={n: p.grad.clone() for n, p in m.named_parameters() if n in keep},
                           state_dict={k: v.clone() for k, v in m.state_dict().items()},
                           cfg=dict(img_size=img, patch_size=16, depth=4, embed_dim=32, decoder_embed_dim=32, decoder_depth=2))
    torch.manual_seed(78)        # initialisation contract
    m = ref.mae_vim.MaskedAutoencoderViM(img_size=64, patch_size=16, depth=2, embed_dim=32, decoder_embed_dim=32,
                                         decoder_depth=1, rms_norm=True, residual_in_fp32=True, fused_add_norm=True,
                                         ssm_cfg={"use_fast_path": False})
    cases["mae_vim_init_seed78"] = dict(state_dict={k: v.clone() for k, v in m.state_dict().items()},
                                        cfg=dict(img_size=64, patch_size=16, depth=2, embed_dim=32, decoder_embed_dim=32,
                                                 decoder_depth=1))
    # ---- ChannelVim, eval mode (HCS off) and one Spatial-First variant
    for name, img, chans, order in (("channelvim_64_c3", 64, 3, "Channel-First"), ("channelvim_32_c5_spatial", 32, 5, "Spatial-First")):
        torch.manual_seed(71)
        model = ref.chan_vim.VisionMamba(img_size=img, patch_size=16, depth=4, embed_dim=32, channels=chans, num_classes=10,
                                         rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                                         if_abs_pos_embed=True, if_cls_token=True, drop_path_rate=0.0, scan_order=order,
                                         ssm_cfg={"use_fast_path": False})
        init_sd = {k: v.clone() for k, v in model.state_dict().items()}
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith(("D", "D_b", "norm.weight", "layernorm.weight", "norm_f.weight")):
                    p.add_(0.2 * torch.randn_like(p))
                elif n.endswith(("layernorm.bias", "head.bias", "patch_embed.proj.bias")):
                    p.add_(0.1 * torch.randn_like(p))
        model.eval()
        x = torch.randn(2, chans, img, img)
        logits = model(x)
        g = torch.randn_like(logits)
        logits.backward(g)
        grads = {n: p.grad.clone() for n, p in model.named_parameters()
                 if n in ("pos_embed", "cls_token", "head.weight", "layers.0.mixer.in_proj.weight", "layers.1.mixer.A_b_log",
                          "layers.3.mixer.x_proj_b.weight", "layers.2.norm.weight", "patch_embed.proj.bias",
                          "patch_embed.proj.weight", "patch_embed.channel_embed.weight", "norm_f.weight",
                          "layers.1.mixer.conv1d.weight", "layers.2.mixer.dt_proj.bias")}
        cases[name] = dict(img=img, channels=chans, scan_order=order, init_seed=71,
                           init_probe={k: init_sd[k] for k in ("cls_token", "layers.0.mixer.in_proj.weight",
                                                               "layers.3.mixer.dt_proj.bias", "pos_embed",
                                                               "patch_embed.channel_embed.weight")},
                           state_dict={k: v.clone() for k, v in model.state_dict().items()},
                           x=x, logits=logits.detach(), g=g, grads=grads,
                           cfg=dict(patch_size=16, depth=4, embed_dim=32, num_classes=10))
    save("baselines.pt", cases)


def gen_channel_variants():
    """Remaining channel-model variants: scan_order="Spatial-First" of the channel model and the
    "2-D compress" model (row-wise -> column-wise -> channel-wise scan cycle).  Tiny backbones, eval mode (HCS off)."""
    cases = {}
    # reference quirk: models_channel_mamba_faster_2dcompress.create_block passes max_tokens_per_patch to a Block whose
    # __init__ does not take it (:363-375 vs :175-190), so the model cannot be constructed as shipped; the keyword is
    # dropped here (it is unused in the sibling model's Block as well)
    _orig_init = ref.chan2.Block.__init__
    if not getattr(_orig_init, "_patched", False):
        def _init(self, *a, max_tokens_per_patch=None, **k):
            _orig_init(self, *a, **k)
        _init._patched = True
        ref.chan2.Block.__init__ = _init
    for name, mod, kw, img, chans, depth in (
            ("spatial_first_64x96_c3", ref.chan, dict(scan_order="Spatial-First", if_abs_pos_embed=True), (64, 96), 3, 4),
            ("compress2d_64x96_c4", ref.chan2, dict(if_abs_pos_embed=True), (64, 96), 4, 6),
            ("compress2d_64x64_c3_nopos", ref.chan2, dict(), (64, 64), 3, 3)):
        torch.manual_seed(41)
        model = mod.VisionMamba(img_size=img, patch_size=16, depth=depth, embed_dim=32, channels=chans, num_classes=10,
                                rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                                drop_path_rate=0.0, **kw)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith(("D", "D_b", "norm.weight", "layernorm.weight", "norm_f.weight")):
                    p.add_(0.2 * torch.randn_like(p))
                elif n.endswith(("layernorm.bias", "head.bias", "patch_embed.proj.bias")):
                    p.add_(0.1 * torch.randn_like(p))
        model.eval()
        x = torch.randn(2, chans, *img)
        logits = model(x)
        g = torch.randn_like(logits)
        logits.backward(g)
        keep = ("pos_embed", "head.weight", "layers.0.mixer.in_proj.weight", "layers.1.mixer.A_b_log",
                "layers.2.mixer.x_proj_b.weight", "layers.2.norm.weight", "patch_embed.proj.bias",
                "patch_embed.proj.weight", "patch_embed.channel_embed.weight", "norm_f.weight",
                "layers.1.mixer.conv1d.weight", "layers.2.mixer.dt_proj.bias", "layers.2.mixer.out_proj.weight",
                "layers.5.mixer.A_log")
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if n in keep and p.grad is not None}
        cases[name] = dict(img=img, channels=chans, depth=depth, kw=kw,
                           state_dict={k: v.clone() for k, v in model.state_dict().items()},
                           x=x, logits=logits.detach(), g=g, grads=grads)
    save("channel_variants.pt", cases)


def gen_config34():
    """BASELINE config 3 (FastVim-B): the whole FastVim-B model at bs = 2 with parameters from the seeded recipe
    (re-derivable on the GPU box, not stored), and a colwise (Pool_row) tiny model.  The FastVim-B mixer on the 14x14
    and 128-column grids is compared with the (golden-pinned) fp64 oracle directly in tests/test_config34_gpu.py."""
    fv = ref.fastvim
    sd = make_state_dict(seed=9, embed_dim=768, depth=24)
    model = fv.vim_base_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(drop_path_rate=0.0)
    model.load_state_dict(sd, strict=True)
    model.eval()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(223))
    logits = model(x)
    g = torch.randn(logits.shape, generator=torch.Generator().manual_seed(224))
    logits.backward(g)
    grads = {n: p.grad.clone() for n, p in model.named_parameters()
             if n in ("head.bias", "norm_f.weight", "layers.0.norm.weight", "layers.23.mixer.D_b",
                      "layers.12.mixer.A_log", "layers.5.mixer.dt_proj.bias", "patch_embed.proj.bias",
                      "layers.7.mixer.conv1d_b.weight", "layers.0.mixer.x_proj.weight", "layers.11.mixer.layernorm.weight")}
    save("model_fastvim_b.pt", dict(param_seed=9, x_seed=223, g_seed=224, x_probe=x[0, 0, :2, :8].clone(),
                                     logits=logits.detach(), grads=grads))

    # colwise scan path (models/fastvim.py:45-51, 97-98): patch grid transposed before flattening
    torch.manual_seed(16)
    img = (64, 64)      # square: the reference's pos-embed resize call is broken when the transposed grid differs
    model = fv.VisionMamba(img_size=img, patch_size=16, depth=4, embed_dim=32, channels=3, num_classes=10,
                           rms_norm=True, residual_in_fp32=True, fused_add_norm=True, final_pool_type="mean",
                           if_abs_pos_embed=True, drop_path_rate=0.0, scanpath_type="colwise")
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith(("D", "D_b", "norm.weight", "layernorm.weight", "norm_f.weight")):
                p.add_(0.2 * torch.randn_like(p))
            elif n.endswith(("layernorm.bias", "head.bias", "patch_embed.proj.bias")):
                p.add_(0.1 * torch.randn_like(p))
    model.eval()
    x = torch.randn(2, 3, *img)
    logits = model(x)
    g = torch.randn_like(logits)
    logits.backward(g)
    grads = {n: p.grad.clone() for n, p in model.named_parameters()
             if n in ("pos_embed", "head.weight", "layers.0.mixer.in_proj.weight", "layers.1.mixer.A_b_log",
                      "layers.3.mixer.x_proj_b.weight", "patch_embed.proj.weight", "layers.1.mixer.conv1d.weight")}
    save("model_colwise.pt", dict(img=img, state_dict={k: v.clone() for k, v in model.state_dict().items()}, x=x,
                                   logits=logits.detach(), g=g, grads=grads))


if __name__ == "__main__":
    which = sys.argv[1:] or ["scan", "compressed_scan", "conv", "norm", "mixer", "model", "channel", "vim", "masked", "mae", "config34", "channel_variants", "baselines"]
    for w in which:
        globals()["gen_" + w]()
