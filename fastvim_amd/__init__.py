"""fastvim_amd -- MI355X (gfx950) native FastVim backbone hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd plumbing,
torch.distributed); every hot op is a hand-written HIP kernel in
``libfastvim_hip.so`` reached through the C ABI in ``include/fastvim_hip.h``.
There is no CPU or eager-PyTorch fallback for those ops: they raise if the
library is missing or the tensors are not on a GPU.
"""
__version__ = "0.1.0"
