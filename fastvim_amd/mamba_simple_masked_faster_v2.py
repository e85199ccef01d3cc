"""``mamba_ssm.modules.mamba_simple_masked_faster_v2.Mamba_masked`` of the reference computes the same function as
``mamba_simple_masked_faster.Mamba_masked`` (``scatter_add_`` along the sequence instead of ``index_add_`` over
flattened rows for the constant-divide row means; same gather, same quirk) -- checked bit for bit on CPU against the
goldens of tests/golden/masked.pt when they were generated.  Same kernels here."""
from .mamba_simple_masked_faster import Mamba_masked, MaskedFastVimMixerFn  # noqa: F401
