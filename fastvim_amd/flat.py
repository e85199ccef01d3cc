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
import re
import warnings

import torch
import torch.distributed as dist

from . import _lib as L
from .layernorm import RMSNorm
from .mamba_simple_faster import Mamba, _GroupedWgrad, _SideStream, flush_wgrads, group_wgrads
from .mixer_ops import defer_reductions, drop_reductions, flush_reductions, pending_reductions

_LAYER_RE = re.compile(r"^((?:.*\.)?[A-Za-z_]*layers\.\d+)\.")      # "layers.3.mixer.A_log" -> "layers.3"

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
        """``comm_dtype``: wire format of the gradient exchange.  None (default) = fp32, what the reference's DDP sums
        (imagenet_classification/train.py:34-43); ``torch.bfloat16`` or ``"auto"`` (bf16 for buckets of >= 100 MB) are
        explicit opt-ins, like the reference's detection-only fp16 hook (detection/vitdet/fp16_compression_hook.py:17-26)."""
        self.group = process_group
        self.comm_dtype = comm_dtype
        self.chunk = max(1, chunk_bytes // 4)
        named = dict(model.named_parameters())
        # Layer-major layout: [parameters before the block stack | block 0 | block 1 | ... | parameters after it], so
        # that the gradient of a run of blocks is ONE contiguous slice -- a bucket of the overlapped gradient exchange
        # (fastvim_amd/ddp.py).  Inside a block the groups of _MIXER_GROUPS come first, each contiguous.
        mixers = {mname: mod for mname, mod in model.named_modules() if isinstance(mod, Mamba)}
        units, unit_of = [], {}          # unit = (key, [names]); key = "<prefix>layers.<i>" for block parameters, else None
        for n, p in named.items():
            if not p.requires_grad:
                continue
            m_ = _LAYER_RE.match(n)
            key = m_.group(1) if m_ else None
            if key is None:
                if not units or units[-1][0] is not None:
                    units.append((None, []))
                units[-1][1].append(n)
            else:
                if key not in unit_of:
                    unit_of[key] = len(units)
                    units.append((key, []))
                units[unit_of[key]][1].append(n)
        order, regions = [], []      # regions: (mixer, group name, [param names])
        self.units = []              # (key, first name, last name) -> element ranges are filled in below
        for key, names in units:
            front = []
            for mname, mod in mixers.items():
                if key is not None and (mname + ".").startswith(key + "."):
                    for gname, keys in _MIXER_GROUPS:
                        full = [f"{mname}.{k}" if mname else k for k in keys]
                        if all(f in named and named[f].requires_grad for f in full):
                            regions.append((mod, gname, full))
                            front += full
            rest = [n for n in names if n not in set(front)]
            self.units.append((key, len(order), len(order) + len(front) + len(rest)))
            order += front + rest
        if not any(k is not None for k, _ in units):      # no block stack found: mixers anywhere in the model
            for mname, mod in mixers.items():
                for gname, keys in _MIXER_GROUPS:
                    full = [f"{mname}.{k}" if mname else k for k in keys]
                    if all(f in named and named[f].requires_grad for f in full):
                        regions.append((mod, gname, full))
            front = [f for _, _, full in regions for f in full]
            order = front + [n for n in order if n not in set(front)]
            self.units = [(None, 0, len(order))]
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
        total = off
        # element range of every unit (block / run of non-block parameters), in layout order
        self.unit_ranges = [(key, offs[i0] if i0 < len(offs) else total, offs[i1] if i1 < len(offs) else total)
                            for key, i0, i1 in self.units if i1 > i0]
        with torch.no_grad():
            for n, p, o in zip(order, params, offs):
                k = p.numel()
                self.param_flat[o:o + k].copy_(p.detach().reshape(-1))
                p.data = self.param_flat[o:o + k].view_as(p)
                p.grad = self.grad_flat[o:o + k].view_as(p)
                p._fv_shadow = self.shadow_flat[o:o + k].view_as(p)
                p._fv_shadow_version = -1
        self._params = params
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
                b = getattr(mod, "bias", None)
                if isinstance(mod, (torch.nn.Linear, torch.nn.Conv2d)) and b is not None and b.requires_grad:
                    b._fv_direct = True              # LinearFn / the patch projection sum their bias gradient in place
        pe = getattr(model, "pos_embed", None)
        if isinstance(pe, torch.nn.Parameter) and pe.requires_grad:
            pe._fv_direct = True                     # (the patch projection's per-token table gradient)
        # transposed bf16 shadows of the in_proj weights the fused conv-adjoint + data-gradient launch streams into MFMA
        # registers (fv_mixer_conv_pool_bwd_dgrad: d_model 192, d_inner 384); re-made by ONE launch after every shadow refresh
        self._t_src, self._t_dst, self._t_params = [], [], []
        if shadow_dtype == torch.bfloat16 and dev.type == "cuda":
            for mod in mixers.values():
                w = mod.in_proj.weight
                if tuple(w.shape) == (768, 192) and w.requires_grad and getattr(w, "_fv_shadow", None) is not None:
                    wt = torch.empty(192, 768, device=dev, dtype=shadow_dtype)
                    w._fv_shadow_t = wt
                    w._fv_shadow_t_version = -1
                    self._t_src.append(w._fv_shadow)
                    self._t_dst.append(wt)
                    self._t_params.append(w)
        # ... and of the x_proj weight pairs of the wide models (d_inner >= 768), (2, W, d_inner) -> (2, d_inner, W): the
        # x_proj adjoint's data half streams them K-contiguous into the bf16 matrix cores (fv_mixer_xproj_bwd3)
        self._tx_src, self._tx_dst, self._tx_params = [], [], []
        if shadow_dtype == torch.bfloat16 and dev.type == "cuda":
            for mod in mixers.values():
                fv = mod.__dict__.get("_fv", {})
                ws = fv.get("Wx2_shadow")
                if ws is None or ws.dtype != torch.bfloat16:
                    continue
                _, Wd, d_in = ws.shape
                if d_in >= 768 and d_in % 64 == 0 and Wd % 8 == 0:
                    wt = torch.empty(2, d_in, Wd, device=dev, dtype=shadow_dtype)
                    fv["Wx2_shadow_t"] = wt
                    fv["Wx2_t_params"] = (mod.x_proj.weight, mod.x_proj_b.weight)
                    for k in range(2):
                        self._tx_src.append(ws[k])
                        self._tx_dst.append(wt[k])
                    for w in fv["Wx2_t_params"]:
                        w._fv_shadow_t_version = -1
                        self._tx_params.append(w)
        self.refresh_shadow()
        # ``model.load_state_dict`` writes the fp32 masters in place: re-cast the shadow right away (the version check in
        # ``_shadow`` would also catch it at the next forward, but a captured HIP graph never runs that check again)
        self._hook = model.register_load_state_dict_post_hook(lambda module, incompatible: self.refresh_shadow())
        # queued gradient work (deferred partial sums, grouped weight-gradient GEMMs) is process-wide state of the
        # kernels' wrappers: remember what it was so ``close()`` can put it back
        from .mixer_ops import _Deferred
        self._saved_switches = (_Deferred.enabled, _GroupedWgrad.enabled, _SideStream.enabled)
        defer_reductions(True)
        group_wgrads(True)
        # (weight-gradient GEMMs on a second stream, joined in finish_backward, measured neutral-to-slower on MI355X under
        # graph replay -- per GEMM 12.35 vs 12.04 ms/step in round 1; as groups of 8 / 16 / 32 problems launched while
        # backward is still running 6.54 / 6.52 / 6.37 vs 6.16 ms in round 2: the MFMA groups and the latency-bound row /
        # scan kernels slow each other down by more than the serial tail costs.  _SideStream stays off and the groups run
        # after the last block unless a caller switches them on)
        _SideStream.enabled = False

    def close(self):
        """Issue whatever is queued and restore the wrappers' process-wide switches (deferred reductions, grouped weight
        gradients, side stream) to what they were before this training state was attached.  Also the exit of
        ``with FlatTrainingState(model) as flat:``."""
        if self._saved_switches is None:
            return
        self.finish_backward()
        d, g, s_ = self._saved_switches
        defer_reductions(d)
        group_wgrads(g)
        _SideStream.enabled = s_
        self._hook.remove()
        self._saved_switches = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ------------------------------------------------------------------ per-step operations
    def zero_grad(self):
        """Zero the flat gradient.  Gradient work still queued from a backward pass that was never finished
        (``finish_backward`` / ``allreduce_mean_`` / ``FlatAdamW.step`` all finish it) would be computed only to be
        zeroed: it is dropped instead, with a warning."""
        _SideStream.join()
        n = len(_GroupedWgrad.jobs) + len(_GroupedWgrad.sums) + pending_reductions()
        if n:
            _GroupedWgrad.jobs = []
            _GroupedWgrad.sums = []
            _GroupedWgrad.rows_jobs = []
            drop_reductions()
            warnings.warn(f"FlatTrainingState.zero_grad(): {n} queued gradient jobs of an unfinished backward pass "
                          "were discarded -- call finish_backward() (or allreduce_mean_() / optimizer.step()) after "
                          "loss.backward()", RuntimeWarning, stacklevel=2)
        self.grad_flat.zero_()

    def finish_backward(self):
        """Join the weight-gradient stream and issue the queued gradient reductions; call after
        loss.backward(), before reading any .grad."""
        flush_wgrads()            # the last group of weight-gradient GEMMs (earlier ones went out during backward), the
                                  # join of the weight-gradient stream, their partial sums queued ...
        flush_reductions()        # ... and issued here

    def refresh_shadow(self):
        """Re-cast every bf16 shadow weight from its fp32 master.  Needed only after writing ``param_flat`` directly
        (in-place writes through the parameters -- load_state_dict, copy_, a torch optimizer -- are detected by
        version counter; the fused optimizer kernel refreshes the shadow itself)."""
        self.shadow_flat.copy_(self.param_flat)
        for p in self._params:
            p._fv_shadow_version = p._version
        self.refresh_transposed()

    def refresh_transposed(self):
        """Re-make the transposed in_proj shadows from ``shadow_flat`` (one launch; called after every shadow refresh:
        here and by the fused optimizer step)."""
        from .mixer_ops import transpose_bf16_batched
        for src, dst, params in ((self._t_src, self._t_dst, self._t_params),
                                 (getattr(self, "_tx_src", []), getattr(self, "_tx_dst", []), getattr(self, "_tx_params", []))):
            if src:
                transpose_bf16_batched(src, dst)
                for p in params:
                    p._fv_shadow_t_version = p._version

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def buckets(self, n_buckets, taper=True):
        """Cut the block stack into ``n_buckets`` runs of whole blocks.  Returns them in the order BACKWARD completes
        them (last blocks first): a list of dicts with ``bounds`` (element range of the flat gradient: the run's blocks
        plus, for the first / last run, the parameters after / before the stack) and ``layers`` ((lo, hi) block indices of
        the run, in forward order).  ``taper``: the runs shrink towards the front of the stack (gradient sizes in the
        ratio n+1 : n : ... : 2 in backward order; 24 blocks in 3 buckets: 10, 8, 6) -- the exchange of the run backward
        reaches last is the one nothing overlaps, so it is the smallest, and the runs exchanged under the rest of
        backward are the large ones (whose weight-gradient groups also fill the chip better).  False: equal sizes."""
        blocks = [(i, u) for i, u in enumerate(self.unit_ranges) if u[0] is not None]
        if not blocks:
            return [{"bounds": (0, self.grad_flat.numel()), "layers": (0, 0)}]
        first_u, last_u = blocks[0][0], blocks[-1][0]
        nb = len(blocks)
        n_buckets = max(1, min(n_buckets, nb))
        sizes = [u[2] - u[1] for _, u in blocks]
        w = [i + 2 if taper else 1 for i in range(n_buckets)]              # forward order: the first run is the smallest
        cum = [sum(sizes) * sum(w[:i + 1]) / sum(w) for i in range(n_buckets)]
        runs, acc, lo = [], 0, 0
        for j, sz in enumerate(sizes):
            acc += sz
            left = nb - 1 - j
            if (len(runs) < n_buckets - 1 and acc >= cum[len(runs)] - 1e-9 and left >= n_buckets - 1 - len(runs)) \
                    or j == nb - 1:
                runs.append((lo, j + 1))
                lo = j + 1
        out = []
        for k, (a, b) in enumerate(runs):
            e0 = self.unit_ranges[first_u + a][1] if k > 0 else 0
            e1 = self.unit_ranges[first_u + b - 1][2] if k < len(runs) - 1 else self.grad_flat.numel()
            out.append({"bounds": (e0, e1), "layers": (a, b)})
        return out[::-1]

    def make_exchange(self, n_buckets=1):
        """The gradient exchange over this state's flat gradient, cut into ``n_buckets`` (fastvim_amd/ddp.py)."""
        from .ddp import GradExchange
        bk = self.buckets(n_buckets)
        self.exchange = GradExchange(self.grad_flat, [b["bounds"] for b in bk], self.group, self.comm_dtype, self.chunk * 4)
        self.exchange.layers = [b["layers"] for b in bk]
        return self.exchange

    def allreduce_mean_(self):
        """Sum the flat gradient across ranks (RCCL) and divide by the world size (torch DDP semantics).  Finishes
        the backward pass first (queued weight-gradient GEMMs and partial sums), so ``backward -> allreduce_mean_ ->
        step`` is a complete sequence."""
        self._allreduce(True)

    def allreduce_sum_(self):
        """The same without the division: follow it with ``FlatAdamW.step(grad_scale=1 / world_size)`` -- the optimizer
        kernel scales the gradient as it reads it, which saves the full pass over the buffer that the mean costs."""
        self._allreduce(False)

    def _allreduce(self, mean):
        self.finish_backward()
        if self.world_size == 1:
            return
        if getattr(self, "exchange", None) is None:
            self.make_exchange(1)
        self.exchange.allreduce_(mean=mean)


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

    # ------------------------------------------------------------------ checkpoint I/O
    def _named_slices(self, buf):
        f = self.flat
        named = dict(zip(f.names, f._params))
        return {n: buf[o:o + named[n].numel()].view_as(named[n]) for n, o in f.offsets.items()}

    def state_dict(self):
        """Portable optimizer state: per-parameter NAMED tensors (no flat offsets / padding), the step count and the
        hyper-parameters -- the information of ``torch.optim.AdamW.state_dict()`` keyed by parameter name."""
        ea, es = self._named_slices(self.exp_avg), self._named_slices(self.exp_avg_sq)
        return {"state": {n: {"exp_avg": ea[n].detach().clone(), "exp_avg_sq": es[n].detach().clone()} for n in ea},
                "step": float(self.step_t.item()), "lr": float(self.lr.item()), "betas": tuple(self.betas),
                "eps": self.eps, "weight_decay": self.weight_decay, "ema_decay": self.ema_decay,
                "ema": None if self.ema is None else {n: v.detach().clone() for n, v in self._named_slices(self.ema).items()}}

    def load_state_dict(self, sd):
        ea, es = self._named_slices(self.exp_avg), self._named_slices(self.exp_avg_sq)
        missing = set(ea) - set(sd["state"])
        if missing:
            raise KeyError(f"FlatAdamW.load_state_dict: no state for {sorted(missing)[:4]} ...")
        with torch.no_grad():
            for n in ea:
                ea[n].copy_(sd["state"][n]["exp_avg"])
                es[n].copy_(sd["state"][n]["exp_avg_sq"])
            self.step_t.fill_(float(sd["step"]))
            self.lr.fill_(float(sd["lr"]))
            if self.ema is not None and sd.get("ema") is not None:
                for n, v in self._named_slices(self.ema).items():
                    v.copy_(sd["ema"][n])
        self.betas, self.eps, self.weight_decay = tuple(sd["betas"]), sd["eps"], sd["weight_decay"]

    def ema_state_dict(self, prefix=""):
        """The EMA weights under the model's own ``state_dict`` key names (``prefix`` + name).  The reference's
        Lightning checkpoints store them UNPREFIXED as ``state_dict_ema`` -- ``get_state_dict(self.ema, unwrap_model)``
        is ``ModelEmaV2.module.state_dict()``, reloaded by ``self.ema.module.load_state_dict``
        (supervised_imagenet.py:107-114) -- and ``MM_FastVim.load_pretrained`` prefers that entry when present."""
        if self.ema is None:
            raise RuntimeError("FlatAdamW was built without ema_decay")
        return {prefix + n: v.detach().clone() for n, v in self._named_slices(self.ema).items()}

    def step(self, grad_scale=1.0):
        """``grad_scale``: factor applied to every gradient element as it is read (``1 / world_size`` after an
        ``allreduce_sum_`` / ``GradExchange.finish(mean=False)``: the data-parallel mean without its own pass)."""
        f = self.flat
        f.finish_backward()         # idempotent: nothing queued in the steady state of a captured step
        rc = L.lib().fv_adamw_flat(
            L.ptr(f.param_flat), L.ptr(f.grad_flat), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), L.ptr(self.ema),
            L.ptr(f.shadow_flat), L.ptr(self.decay_mask), L.ptr(self.lr), L.ptr(self.step_t),
            ctypes.c_float(self.betas[0]), ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps),
            ctypes.c_float(self.weight_decay), ctypes.c_float(self.ema_decay), ctypes.c_float(grad_scale),
            ctypes.c_size_t(f.param_flat.numel()), L.stream_of(f.param_flat))
        L.check(rc, "adamw_flat")
        f.refresh_transposed()      # (the kernel above has just re-cast the bf16 shadows)


def save_checkpoint(path, model, opt, prefix="backbone.", **extra):
    """Write a checkpoint in the layout of the reference's Lightning checkpoints: ``state_dict`` (model keys behind
    ``prefix``: the LightningModule holds the model as ``self.backbone``), ``state_dict_ema`` (when the optimizer tracks an
    EMA; UNPREFIXED model keys, as ``on_save_checkpoint`` writes it, supervised_imagenet.py:107-110) and the optimizer
    state -- loadable by the reference's ``on_load_checkpoint``, ``MM_FastVim.load_pretrained`` and ``load_checkpoint``."""
    ck = {"state_dict": {prefix + k: v.detach().clone() for k, v in model.state_dict().items()},
          "optimizer_states": [opt.state_dict()], **extra}
    if opt.ema is not None:
        ck["state_dict_ema"] = opt.ema_state_dict("")
    torch.save(ck, path)
    return ck


def load_checkpoint(path_or_dict, model, opt=None, prefix="backbone.", use_ema=False):
    """Resume: model weights (or the EMA weights) and, when ``opt`` is given, the optimizer state.  The flat training
    state's bf16 shadow follows through the load_state_dict hook."""
    ck = torch.load(path_or_dict, map_location="cpu", weights_only=False) if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
    sd = ck["state_dict_ema"] if (use_ema and "state_dict_ema" in ck) else ck["state_dict"]
    # the prefix is stripped where it is present (``key.replace("backbone.", "")``, models/fastvim.py:617): ``state_dict``
    # carries it, the reference's ``state_dict_ema`` does not
    model.load_state_dict({(k[len(prefix):] if prefix and k.startswith(prefix) else k): v for k, v in sd.items()}, strict=True)
    if opt is not None:
        opt.load_state_dict(ck["optimizer_states"][0])
    return ck
