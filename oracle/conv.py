"""Causal depthwise conv1d (+SiLU) oracle (test infrastructure; see oracle/__init__.py).

The CUDA op lives in the un-vendored PyPI package ``causal-conv1d==1.1.3.post1``
(reference README.md:43); its published contract is
``y[b,d,t] = act(bias[d] + sum_{k<W} w[d,k] * x[b,d,t-(W-1)+k])`` with zero left
padding -- the same formula as the reference's in-tree fallback
``act(conv1d(x)[..., :L])`` with ``padding=W-1, groups=D``
(mamba-1p1p1/mamba_ssm/modules/mamba_simple.py:302-303).  Call sites:
mamba_simple_faster.py:274-285.
"""
import torch
import torch.nn.functional as F


def causal_conv1d_oracle(x, weight, bias=None, activation=None, anticausal=False,
                         compute_dtype=torch.float64, out_dtype=None):
    """x: (B,D,L); weight: (D,W); bias: (D).  Written as an explicit shift-and-add.

    ``anticausal=True`` evaluates ``y[t] = act(bias + sum_k w[k] * x[t+(W-1)-k])``,
    which equals ``flip(causal_conv(flip(x)))`` -- the FastVim backward direction
    (mamba_simple_faster.py:272,280-285) without materialising the flips."""
    assert activation in (None, "silu", "swish")
    out_dtype = x.dtype if out_dtype is None else out_dtype
    cd = compute_dtype
    Bsz, D, L = x.shape
    W = weight.shape[1]
    xf, wf = x.to(cd), weight.to(cd)
    acc = torch.zeros(Bsz, D, L, dtype=cd, device=x.device)
    if bias is not None:
        acc = acc + bias.to(cd)[None, :, None]
    for k in range(W):
        s = (W - 1) - k  # distance into the past (causal) / future (anticausal)
        if s >= L:
            continue
        if not anticausal:
            shifted = F.pad(xf[:, :, : L - s], (s, 0))
        else:
            shifted = F.pad(xf[:, :, s:], (0, s))
        acc = acc + wf[None, :, k, None] * shifted
    y = F.silu(acc) if activation is not None else acc
    return y.to(out_dtype)
