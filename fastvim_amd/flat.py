"""Flat training state: parameters, gradients and bf16 shadow weights in three contiguous buffers.

MI355X-first replacement for per-parameter autograd bookkeeping (the reference relies on torch DDP
buckets + autocast casts + AccumulateGrad nodes, imagenet_classification/train.py:34-43):

* ``param_flat`` (fp32 masters; every ``p.data`` is a view), ``grad_flat`` (fp32; every ``p.grad`` is
  a view -> ONE RCCL all-reduce region, ONE zero kernel), ``shadow_flat`` (bf16 copy refreshed by ONE
  cast kernel after the optimizer step; the GEMMs read it instead of casting 6 weights per block
  per step);
* parameters of each mixer are laid out so that the per-block partial sums produced by the fused
  backward kernels (csrc/mixer_bwd.hip, csrc/scan_cl.hip) land in ONE contiguous gradient region
  each: the reduction kernel accumulates straight into ``grad_flat`` and autograd never sees those
  gradients (no AccumulateGrad launches, no strided adds).
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib as L
from .layernorm import RMSNorm
from .mamba_simple_faster import Mamba, _SideStream, flush_wgrads, group_wgrads
from .mixer_ops import defer_reductions, flush_reductions

# per-mixer parameter order; each group is one contiguous region matching a kernel's partial layout
_MIXER_GROUPS = (
    ("conv", ("conv1d.weight", "conv1d_b.weight", "conv1d.bias", "conv1d_b.bias", "D", "D_b")),
    ("scan", ("A_log", "dt_proj.weight", "dt_proj.bias", "A_b_log", "dt_proj_b.weight", "dt_proj_b.bias")),
    ("ln", ("layernorm.weight", "layernorm.bias")),
    ("xproj", ("x_proj.weight", "x_proj_b.weight")),
)


class FlatTrainingState:
    def __init__(self, model, shadow_dtype=torch.bfloat16, process_group=None, comm_dtype=None,
                 chunk_bytes=256 << 20):
        self.group = process_group
        self.comm_dtype = comm_dtype
        self.chunk = max(1, chunk_bytes // 4)
        named = dict(model.named_parameters())
        order, seen = [], set()
        regions = []      # (mixer, group name, [param names])
        for mname, mod in model.named_modules():
            if not isinstance(mod, Mamba):
                continue
            for gname, keys in _MIXER_GROUPS:
                full = [f"{mname}.{k}" if mname else k for k in keys]
                if all(f in named and named[f].requires_grad for f in full):
                    regions.append((mod, gname, full))
                    for f in full:
                        order.append(f)
                        seen.add(f)
        for n, p in named.items():
            if n not in seen and p.requires_grad:
                order.append(n)
        self.names = order
        params = [named[n] for n in order]
        dev = params[0].device
        # offsets in multiples of 8 elements: every fp32 AND bf16 view is 16-byte aligned (MFMA GEMM loads)
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + 7) // 8 * 8
        self.param_flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad_flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.shadow_flat = torch.zeros(off, device=dev, dtype=shadow_dtype)
        self.offsets = dict(zip(order, offs))
        with torch.no_grad():
            for n, p, o in zip(order, params, offs):
                k = p.numel()
                self.param_flat[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.param_flat[o:o + k].view_as(p)
                p.grad = self.grad_flat[o:o + k].view_as(p)
                p._fv_shadow = self.shadow_flat[o:o + k].view_as(p)
        for mod, gname, full in regions:
            lo = self.offsets[full[0]]
            last = named[full[-1]]
            hi = self.offsets[full[-1]] + last.numel()
            contiguous = all(self.offsets[a] + named[a].numel() == self.offsets[b]
                             for a, b in zip(full[:-1], full[1:]))
            if not contiguous:            # padding slipped in (numel % 4 != 0): fall back to autograd for this group
                continue
            fv = mod.__dict__.setdefault("_fv", {})
            fv[gname + "_grad"] = self.grad_flat[lo:hi]
            if gname == "xproj":
                shp = (2,) + tuple(named[full[0]].shape)
                fv["Wx2"] = self.param_flat[lo:hi].view(shp)
                fv["Wx2_shadow"] = self.shadow_flat[lo:hi].view(shp)
                fv["Wx2_grad"] = self.grad_flat[lo:hi].view(shp)
        for mod in model.modules():
            if isinstance(mod, (RMSNorm, torch.nn.LayerNorm, torch.nn.Linear, torch.nn.Conv2d)) and getattr(mod, "weight", None) is not None:
                if mod.weight.requires_grad:
                    mod.weight._fv_direct = True     # my backward kernels may accumulate into .grad directly
        self.refresh_shadow()
        defer_reductions(True)
        group_wgrads(True)
        # weight-gradient GEMMs on a second stream (joined in finish_backward): measured neutral-to-slower
        # on MI355X under graph replay (12.35 vs 12.04 ms/step), so opt-in only
        _SideStream.enabled = os.environ.get("FASTVIM_WGRAD_STREAM", "0") == "1"

    # ------------------------------------------------------------------ per-step operations
    def zero_grad(self):
        _SideStream.join()
        flush_wgrads()
        flush_reductions()
        self.grad_flat.zero_()

    def finish_backward(self):
        """Join the weight-gradient stream and issue the queued gradient reductions; call after
        loss.backward(), before reading any .grad."""
        _SideStream.join()
        flush_wgrads()            # grouped weight-gradient GEMMs queue their partial sums ...
        flush_reductions()        # ... which are issued here

    def refresh_shadow(self):
        self.shadow_flat.copy_(self.param_flat)

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def allreduce_mean_(self):
        """Sum the flat gradient across ranks (RCCL) and divide by the world size (torch DDP semantics)."""
        _SideStream.join()
        flush_reductions()
        ws = self.world_size
        if ws == 1:
            return
        n = self.grad_flat.numel()
        for s in range(0, n, self.chunk):
            view = self.grad_flat[s:min(n, s + self.chunk)]
            if self.comm_dtype is not None and self.comm_dtype != torch.float32:
                buf = view.to(self.comm_dtype)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                view.copy_(buf)
            else:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        self.grad_flat.div_(ws)


class FlatAdamW:
    """torch.optim.AdamW semantics as ONE HIP kernel over a FlatTrainingState (csrc/optim.hip): AdamW
    update, optional EMA of the weights (timm ModelEmaV2) and the bf16 shadow refresh in a single pass.
    ``no_decay`` is a set of parameter names excluded from weight decay (reference recipe:
    imagenet_classification/utils.py:52-69).  ``lr`` may be changed between steps with ``set_lr`` --
    it lives in device memory, so a captured HIP graph replays with the new value."""

    def __init__(self, flat, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, no_decay=(),
                 ema_decay=None):
        self.flat = flat
        dev = flat.param_flat.device
        n = flat.param_flat.numel()
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        self.ema = flat.param_flat.clone() if ema_decay is not None else None
        self.ema_decay = float(ema_decay or 0.0)
        mask = torch.zeros(n, dtype=torch.uint8)
        named = dict(model.named_parameters())
        no_decay = set(no_decay)
        for name, off in flat.offsets.items():
            p = named[name]
            if name not in no_decay:
                mask[off:off + p.numel()] = 1
        self.decay_mask = mask.to(dev)
        self.lr = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
        self.step_t = torch.zeros(1, device=dev, dtype=torch.float32)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay

    def set_lr(self, lr):
        self.lr.fill_(float(lr))

    def step(self):
        f = self.flat
        rc = L.lib().fv_adamw_flat(
            L.ptr(f.param_flat), L.ptr(f.grad_flat), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), L.ptr(self.ema),
            L.ptr(f.shadow_flat), L.ptr(self.decay_mask), L.ptr(self.lr), L.ptr(self.step_t),
            ctypes.c_float(self.betas[0]), ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps),
            ctypes.c_float(self.weight_decay), ctypes.c_float(self.ema_decay), ctypes.c_size_t(f.param_flat.numel()),
            L.stream_of(f.param_flat))
        L.check(rc, "adamw_flat")
