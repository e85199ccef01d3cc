import sys, torch
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R+'/tests')
from conftest import load_golden
from fastvim_amd import mixer_ops as M
from oracle import fastvim_mixer_oracle, make_state_dict
import torch.nn.functional as F

def run(case, transposed=False, dtype=torch.float32):
    c = load_golden("mixer.pt")[case]
    if "state_dict" in c: sd = c["state_dict"]
    else:
        r = c["param_recipe"]; full = make_state_dict(seed=r["seed"], embed_dim=r["embed_dim"], depth=r["depth"])
        sd = {k[len(r["prefix"]):]: v for k, v in full.items() if k.startswith(r["prefix"])}
    rows, cols = c["token_size"]
    h = c["hidden"]
    Bsz, Ltok, d = h.shape
    ref = c["out"]
    if transposed:
        # feed the memory-order (un-transposed) tokens; mixer sees grid (rows, cols) as the transpose of memory grid (cols, rows)
        hm = h.reshape(Bsz, rows, cols, d).transpose(1, 2).reshape(Bsz, Ltok, d)   # memory grid is (cols, rows)
        ref = ref.reshape(Bsz, rows, cols, d).transpose(1, 2).reshape(Bsz, Ltok, d)
    else:
        hm = h
    p = {k: v.cuda() for k, v in sd.items()}
    hm = hm.cuda()
    d_in = p["in_proj.weight"].shape[0] // 2
    R = p["dt_proj.weight"].shape[1]; N = 16
    xz = (hm.to(dtype) @ p["in_proj.weight"].to(dtype).t()).contiguous()
    cw = p["conv1d.weight"].reshape(d_in, -1).contiguous(); cwb = p["conv1d_b.weight"].reshape(d_in, -1).contiguous()
    xc = M.conv_pool_fwd(xz, cw, p["conv1d.bias"], cwb, p["conv1d_b.bias"], rows, cols, transposed, 0, 1.0)
    Wx = torch.stack([p["x_proj.weight"], p["x_proj_b.weight"]]).to(dtype)
    x_dbl = torch.bmm(xc.reshape(2, Bsz * rows, d_in), Wx.transpose(1, 2)).contiguous()
    yc = M.scan_fwd(xc, x_dbl, p["dt_proj.weight"], p["dt_proj.bias"], p["A_log"], p["dt_proj_b.weight"], p["dt_proj_b.bias"], p["A_b_log"])
    g, _xh, mean, rstd = M.combine_fwd(xz, yc, cw, p["conv1d.bias"], cwb, p["conv1d_b.bias"], p["D"], p["D_b"],
                                  p["layernorm.weight"], p["layernorm.bias"], 1e-5, rows, cols, transposed)
    y = g @ p["out_proj.weight"].to(dtype).t()
    err = (y.float().cpu() - ref).abs().max().item()
    print(case, "transposed" if transposed else "natural", dtype, "max err", err, "max ref", ref.abs().max().item())

for case in ["d32_4x4", "d32_3x5", "d192_14x14"]:
    for tr in (False, True):
        run(case, tr)
run("d192_14x14", False, torch.bfloat16)
run("d192_14x14", True, torch.bfloat16)
