"""CPU checks of the host-side module mirror: constructor surface, state_dict contract, factory names."""
import torch

from oracle.model import state_dict_shapes


def test_state_dict_contract_matches_reference_keys():
    from fastvim_amd.fastvim import FastVimT
    m = FastVimT()
    sd = m.state_dict()
    shapes = state_dict_shapes(embed_dim=192, depth=24)
    assert set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert len(sd) == 462                                           # 462 tensors
    assert sum(p.numel() for p in m.parameters()) == 7166056       # 7.166 M (reference README)
    assert m.no_weight_decay() == {"pos_embed"}
    mix = m.layers[0].mixer
    assert mix.A_log._no_weight_decay and mix.D._no_weight_decay and mix.dt_proj.bias._no_reinit
    assert m.layers[1].mixer.num_of_rows == 14 and m.layers[2].drop_path.drop_prob > 0
    assert isinstance(m.layers[1].drop_path, torch.nn.Identity)   # inter_dpr[1] == dpr[0] == 0


def test_param_counts_small_base():
    from fastvim_amd.fastvim import FastVimB, FastVimS
    assert abs(sum(p.numel() for p in FastVimS().parameters()) - 25.83e6) < 0.01e6
    nb = sum(p.numel() for p in FastVimB().parameters())
    assert abs(nb - 97.67e6) < 0.01e6


def test_droppath_schedule_off_by_one():
    """layer i uses inter_dpr[i] = ([0] + linspace(0, dpr, depth))[i]  (models/fastvim.py:415-433)."""
    from fastvim_amd.fastvim import DropPath, VisionMamba
    m = VisionMamba(img_size=32, depth=4, embed_dim=32, drop_path_rate=0.3, fused_add_norm=True,
                    residual_in_fp32=True, final_pool_type="mean")
    rates = [l.drop_path.drop_prob if isinstance(l.drop_path, DropPath) else 0.0 for l in m.layers]
    exp = [0.0] + torch.linspace(0, 0.3, 4).tolist()
    assert all(abs(a - b) < 1e-7 for a, b in zip(rates, exp[:4]))
    assert m.layers[1].mixer.num_of_rows == 2 and m.layers[0].mixer.num_of_col == 2
