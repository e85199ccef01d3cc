"""Data-parallel gradient exchange for the FastVim training step.

The reference has no communication code of its own: Lightning wraps the model in torch DDP over
NCCL (imagenet_classification/train.py:34-43).  Here it is explicit and minimal: one process per
GPU, gradients live in ONE flat fp32 buffer (``p.grad`` are views), and after backward the buffer
is summed across ranks with RCCL (``torch.distributed`` backend "nccl" on ROCm) in a few large
chunks -- xGMI rings are per-link bound, so few large messages beat torch DDP's 25 MB bucket
default for a 28.7 MB (FastVim-T) .. 391 MB (FastVim-B) gradient -- and averaged.
No BatchNorm exists in FastVim, so nothing else is exchanged.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, params, process_group=None, chunk_bytes=256 << 20, comm_dtype=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.comm_dtype = comm_dtype                     # e.g. torch.bfloat16 to halve xGMI bytes
        numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(numel, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)   # autograd accumulates in place into the view
            off += n
        self.chunk = max(1, chunk_bytes // 4)

    def zero_(self):
        self.flat.zero_()

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def allreduce_mean_(self):
        """Sum the flat gradient across ranks and divide by the world size (torch DDP semantics)."""
        ws = self.world_size
        if ws == 1:
            return
        n = self.flat.numel()
        for s in range(0, n, self.chunk):
            view = self.flat[s:min(n, s + self.chunk)]
            if self.comm_dtype is not None and self.comm_dtype != torch.float32:
                buf = view.to(self.comm_dtype)
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                view.copy_(buf)
            else:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(ws)


def shard_batch(global_batch, rank, world_size):
    """Even split of a global batch; BASELINE configs give the per-GPU batch directly (weak scaling)."""
    assert global_batch % world_size == 0
    per = global_batch // world_size
    return rank * per, (rank + 1) * per
