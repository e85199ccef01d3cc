"""FastChannelVim "2-D compress" backbone: mirror of
models/channel_wise_tokenization/models_channel_mamba_faster_2dcompress.py -- the channel model whose blocks cycle
row-wise scan -> column-wise scan -> channel-wise scan (``Block`` :265-300, ``create_block`` :333-339) on the
2-D compress mixer (mamba_ssm/modules/mamba_simple_channel_faster_2dcompress.py).  Same constructor kwargs
(``if_abs_pos_embed`` defaults to False there, ``use_middle_cls_token`` is accepted and unused, as in the reference),
``state_dict`` keys and entry point.  Channel-First tokens; the column-scan layers (``(layer_idx + 2) % 3 == 0``)
transpose the cell grid physically, like the reference.
"""
from functools import partial

from . import models_channel_mamba_faster as _base
from .mamba_simple_channel_faster_2dcompress import Mamba


def _rotated(layer_idx):
    return (layer_idx + 2) % 3 == 0          # rowwise scan -> colwise scan -> channel wise scan -> repeat


class Block(_base.Block):
    def _rotated(self):
        return self.rotate_every_block is True and _rotated(self.layer_idx)

    def _physical_transpose(self):
        return True


create_block = partial(_base.create_block, mixer_type=Mamba, block_type=Block, rotated=_rotated)


class VisionMamba(_base.VisionMamba):
    _create_block = staticmethod(create_block)

    def __init__(self, *args, if_abs_pos_embed=False, use_middle_cls_token=True, scan_order="Channel-First", **kwargs):
        if scan_order != "Channel-First":
            raise NotImplementedError("2-D compress model: the reference implements scan_order='Channel-First' only")
        super().__init__(*args, if_abs_pos_embed=if_abs_pos_embed, scan_order=scan_order, **kwargs)


def channelvim_small_patch16_224_final_pool_mean_abs_pos_embed_with_noclstok_div2(
        pretrained=False, patch_size=16, stride=16, if_abs_pos_embed=True, **kwargs):
    """models_channel_mamba_faster_2dcompress.py:654-676."""
    if pretrained:
        raise RuntimeError("no pretrained FastChannelVim weights are published (reference url is 'to.do')")
    model = VisionMamba(patch_size=patch_size, stride=stride, if_abs_pos_embed=if_abs_pos_embed, embed_dim=384,
                        depth=24, rms_norm=True, residual_in_fp32=True, fused_add_norm=True, **kwargs)
    model.default_cfg = {}
    return model
