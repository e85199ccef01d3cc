"""fastvim_amd -- MI355X (gfx950) native FastVim backbone hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd plumbing,
torch.distributed); every hot op is a hand-written HIP kernel in
``libfastvim_hip.so`` reached through the C ABI in ``include/fastvim_hip.h``.
There is no CPU or eager-PyTorch fallback for those ops: they raise if the
library is missing or the tensors are not on a GPU.
"""
import os as _os

# ROCm 7.2's HIP-graph "packet capture" path (pre-recorded AQL packets) replays some captured training steps wrongly
# (FastVim-T at 512 px bs=32 and the MAE step at bs >= 64 turn non-finite after a few replays).
# With the switch off kernel nodes are launched one by one, every configuration replays bitwise equal to eager and at
# the same speed.  The runtime reads it when it initialises (the first HIP call), so it is set on import -- import
# fastvim_amd before anything touches the GPU, or export it yourself; ``graph_capture_safe()`` tells which happened.
def _hip_initialised():
    import sys
    torch = sys.modules.get("torch")
    return bool(torch is not None and torch.cuda.is_initialized())


_SWITCH = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
# the setting can only be trusted if it was already exported, or if the runtime had not started when it was set here
_CAPTURE_SAFE = _os.environ.get(_SWITCH) == "0" or (_SWITCH not in _os.environ and not _hip_initialised())
_os.environ.setdefault(_SWITCH, "0")

__version__ = "0.2.0"


def graph_capture_safe():
    """True when HIP-graph capture of the training step is known to replay correctly in this process: the
    packet-capture switch is off and was set before the HIP runtime initialised."""
    return _CAPTURE_SAFE and _os.environ.get(_SWITCH) == "0"
