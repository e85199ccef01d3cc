"""GPU behaviour of the flat training state around its edges: weights loaded after it is attached, checkpoint
resume of the fused optimizer, gradient accumulation over two backward passes, backward -> all-reduce -> step
without an explicit finish_backward()."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(depth=2, dim=192, img=224, classes=50):
    from fastvim_amd.fastvim import VisionMamba
    return VisionMamba(img_size=img, depth=depth, embed_dim=dim, num_classes=classes, rms_norm=True, residual_in_fp32=True,
                       fused_add_norm=True, final_pool_type="mean", if_abs_pos_embed=True, drop_path_rate=0.0).cuda().train()


def _nd(m):
    return {n for n, p in m.named_parameters() if p.ndim <= 1 or n.endswith(".bias") or n in m.no_weight_decay()
            or getattr(p, "_no_weight_decay", False)}


def test_load_state_dict_after_flat_attach_gives_fresh_model_logits():
    """the bf16 shadow weights the GEMMs read must follow ``model.load_state_dict`` (and any other in-place
    write to a parameter) made AFTER FlatTrainingState was attached."""
    from fastvim_amd.flat import FlatTrainingState
    torch.manual_seed(0)
    src = _model()
    sd = {k: v.detach().clone() for k, v in src.state_dict().items()}
    torch.manual_seed(1)
    m = _model()
    x = torch.randn(8, 3, 224, 224, device="cuda")
    with FlatTrainingState(m):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            before = m(x)
        m.load_state_dict(sd)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            after = m(x)
            ref = src(x)
        assert torch.equal(after, ref) and not torch.equal(before, ref)
        # an EMA copy-in / torch optimizer style write: detected by version counter at the next forward
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(0.5)
            for p in src.parameters():
                p.mul_(0.5)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            assert torch.equal(m(x), src(x))


def test_checkpoint_resume_gives_identical_next_steps(tmp_path):
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState, load_checkpoint, save_checkpoint
    torch.manual_seed(0)
    m1 = _model()       # (not a deepcopy: copy.deepcopy drops the parameters' _no_weight_decay attribute)
    x = torch.randn(8, 3, 224, 224, device="cuda")
    g = torch.randn(8, 50, device="cuda")

    def step(m, flat, opt):
        flat.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        (y.float() * g).sum().backward()
        opt.step()                            # finishes the backward pass itself

    f1 = FlatTrainingState(m1)
    o1 = FlatAdamW(f1, m1, lr=1e-3, weight_decay=0.05, no_decay=_nd(m1), ema_decay=0.99)
    for _ in range(3):
        step(m1, f1, o1)
    path = str(tmp_path / "resume.ckpt")
    save_checkpoint(path, m1, o1)
    for _ in range(2):
        step(m1, f1, o1)
    torch.manual_seed(5)
    m2 = _model()                              # different init: everything must come from the checkpoint
    f2 = FlatTrainingState(m2)
    o2 = FlatAdamW(f2, m2, lr=7.0, weight_decay=0.05, no_decay=_nd(m2), ema_decay=0.99)
    load_checkpoint(path, m2, o2)
    for _ in range(2):
        step(m2, f2, o2)
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for n in p1:
        assert torch.equal(p1[n], p2[n]), n
    e1, e2 = o1.ema_state_dict(), o2.ema_state_dict()
    assert all(torch.equal(e1[k], e2[k]) for k in e1)
    f1.close(); f2.close()


def test_gradient_accumulation_two_backwards_one_finish():
    """two backward passes before one finish_backward() queue two partial sums / two grouped weight-gradient
    problems for the SAME gradient; they must be applied one after the other (deterministic), and equal the gradient
    of the summed loss."""
    from fastvim_amd.flat import FlatTrainingState
    torch.manual_seed(0)
    base = _model()
    xs = [torch.randn(16, 3, 224, 224, device="cuda") for _ in range(2)]
    g = torch.randn(16, 50, device="cuda")

    def loss(m, x):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return (m(x).float() * g).sum()

    res = []
    for _ in range(2):
        m = copy.deepcopy(base)
        with FlatTrainingState(m) as flat:
            flat.zero_grad()
            loss(m, xs[0]).backward()
            loss(m, xs[1]).backward()
            flat.finish_backward()
            torch.cuda.synchronize()
            res.append(flat.grad_flat.clone())
    assert torch.equal(res[0], res[1])
    m = copy.deepcopy(base)
    with FlatTrainingState(m) as flat:
        flat.zero_grad()
        (loss(m, xs[0]) + loss(m, xs[1])).backward()
        flat.finish_backward()
        ref = flat.grad_flat.clone()
        names, offs = flat.names, flat.offsets
        params = dict(m.named_parameters())
    for n in names:
        o, k = offs[n], params[n].numel()
        a, b = res[0][o:o + k], ref[o:o + k]
        assert (a - b).abs().max().item() <= 2e-3 * max(1e-3, b.abs().max().item()), n


def test_backward_allreduce_step_without_finish_backward():
    """backward -> allreduce_mean_() -> step() is complete on its own: the projection weight gradients queued for the
    grouped launches reach the optimizer (they used to be flushed by the NEXT zero_grad and discarded)."""
    from fastvim_amd.flat import FlatAdamW, FlatTrainingState
    torch.manual_seed(0)
    base = _model()
    x = torch.randn(16, 3, 224, 224, device="cuda")
    g = torch.randn(16, 50, device="cuda")
    outs = []
    for explicit in (True, False):
        m = copy.deepcopy(base)
        with FlatTrainingState(m) as flat:
            opt = FlatAdamW(flat, m, lr=1e-3, weight_decay=0.05, no_decay=_nd(m))
            flat.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = m(x)
            (y.float() * g).sum().backward()
            if explicit:
                flat.finish_backward()
            flat.allreduce_mean_()
            assert m.layers[0].mixer.in_proj.weight.grad.abs().max().item() > 0
            opt.step()
            outs.append(flat.param_flat.clone())
    assert torch.equal(outs[0], outs[1])


def test_non_square_grid_flat_state_step_with_folded_xproj():
    """on a 224 x 256 px grid (14 x 16 patches) the even layers pool to 14 rows and the rotated ones to 16, so the
    queue of one backward pass holds d x_dbl row jobs of TWO shapes; they used to meet an all-one-shape assert in
    ``chunk_rows_bf16``.  The flat-state step must run (chained blocks, x_proj adjoint folded into the scan backward) and give
    the gradients of the same model stepped without the flat state."""
    from fastvim_amd.flat import FlatTrainingState
    torch.manual_seed(0)
    base = _model(depth=4, img=(224, 256))
    x = torch.randn(32, 3, 224, 256, device="cuda")       # batch a multiple of 32: the grouped x_proj weight gradient path
    g = torch.randn(32, 50, device="cuda")

    def loss(m):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return (m(x).float() * g).sum()

    m = copy.deepcopy(base)
    with FlatTrainingState(m) as flat:
        flat.zero_grad()
        loss(m).backward()
        flat.finish_backward()
        torch.cuda.synchronize()
        got = flat.grad_flat.clone()
        names, offs = flat.names, flat.offsets
    ref_m = copy.deepcopy(base)
    loss(ref_m).backward()
    params = dict(ref_m.named_parameters())
    assert torch.isfinite(got).all()
    for n in names:
        o, k = offs[n], params[n].numel()
        a, b = got[o:o + k], params[n].grad.reshape(-1).float()
        assert (a - b).abs().max().item() <= 5e-3 * max(1e-3, b.abs().max().item()), n
