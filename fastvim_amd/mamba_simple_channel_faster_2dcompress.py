"""FastChannelVim "2-D compress" mixer: mirror of ``Mamba`` in
mamba-1p1p1/mamba_ssm/modules/mamba_simple_channel_faster_2dcompress.py:24-436 -- same constructor kwargs, parameter
names / shapes / initialisation and ``forward(hidden_states (B, L, D), tokens_per_patch) -> (B, L, D)``.

Channel-First token order ``(row, col, channel)``; the pooling axis cycles with the layer (:226-249, 322-330):

* layers with ``(layer_idx + 1) % 3 == 0`` scan over CHANNELS: every (row, col) cell is pooled away, the scan has
  ``tokens_per_patch`` steps and its output is tiled back over the cells -- the fused kernels' geometry
  ``1 x (rows*cols)`` cells of ``tokens_per_patch`` tokens;
* the other layers scan over rows (the model's ``Block`` transposes the cells first on the column-scan layers,
  models_channel_mamba_faster_2dcompress.py:265-300): cols AND channels are pooled away, ``rows`` steps -- the plain
  FastVim geometry ``rows x (cols*tokens_per_patch)``.

The reference implements ``scan_order="Channel-First"`` only (Spatial-First prints "not implemented yet").
"""
from .mamba_simple_channel_faster import Mamba as _ChannelMamba


class Mamba(_ChannelMamba):
    def __init__(self, *args, scan_order="Channel-First", **kwargs):
        if scan_order != "Channel-First":
            raise NotImplementedError("2-D compress mixer: the reference implements scan_order='Channel-First' only")
        super().__init__(*args, scan_order=scan_order, **kwargs)

    def _geometry(self, tokens_per_patch):
        if (self.layer_idx + 1) % 3 == 0:                                     # channel-wise scan
            return 1, self.num_of_rows * self.num_of_col, tokens_per_patch
        return self.num_of_rows, self.num_of_col * tokens_per_patch, 1        # row- / column-wise scan

    def forward(self, hidden_states, tokens_per_patch, inference_params=None, transposed_grid=False):
        if transposed_grid:
            raise RuntimeError("2-D compress mixers take physically transposed tokens (their Block does that)")
        return super().forward(hidden_states, tokens_per_patch, inference_params=inference_params)
