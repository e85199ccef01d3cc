"""CPU oracle for the FastVim backbone hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32/fp64) restatement of the reference
algorithms on the hot path named by BASELINE.json.  It is the *checker*:

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg may import it;
* nothing under ``fastvim_amd/`` (the product) imports it, and the product has
  no CPU fallback -- it raises if the HIP library is missing.

Parity status: **pinned**.  Every function here is checked against outputs of
the reference itself (imported on CPU in the build container through
``tests/golden/_ref_import.py``) frozen as fixtures under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks the oracle against those fixtures on
every CPU test run.  The one third-party op whose source is not in the
reference tree -- ``causal-conv1d==1.1.3.post1`` (reference README.md:43) -- is
pinned to the ``F.conv1d`` formula the reference itself uses as its fallback
(mamba-1p1p1/mamba_ssm/modules/mamba_simple.py:302-303).

Each function cites the reference file:line it follows (paths relative to the
reference root).
"""
from .scan import selective_scan_oracle, selective_scan_ref_port  # noqa: F401
from .conv import causal_conv1d_oracle  # noqa: F401
from .norm import fused_add_norm_oracle  # noqa: F401
from .losses import soft_target_ce_oracle  # noqa: F401
from .mixer import fastvim_mixer_oracle, masked_mixer_oracle  # noqa: F401
from .model import (fastvim_block_oracle, fastvim_forward_oracle, make_state_dict,  # noqa: F401
                    channel_forward_oracle, channel_block_oracle, make_channel_state_dict,
                    vim_forward_oracle, vim_mixer_oracle, mae_forward_oracle, mae_random_masking_oracle)
