"""CPU restatement of the training loss (TEST INFRASTRUCTURE ONLY, like everything under oracle/).

timm.loss.SoftTargetCrossEntropy (timm is a third-party dependency of the reference, not vendored in /root/reference;
used at imagenet_classification/supervised_imagenet.py:83, 109-115): ``torch.sum(-target * F.log_softmax(x, dim=-1),
dim=-1).mean()``.  Parity is pinned on that published formula, evaluated in fp64.
"""
import torch
import torch.nn.functional as F


def soft_target_ce_oracle(x, target):
    """x, target (B, C) -> (loss scalar fp64, d loss / d x (B, C) fp64)."""
    x = x.detach().double().cpu().requires_grad_(True)
    t = target.detach().double().cpu()
    loss = torch.sum(-t * F.log_softmax(x, dim=-1), dim=-1).mean()
    (g,) = torch.autograd.grad(loss, x)
    return loss.detach(), g
