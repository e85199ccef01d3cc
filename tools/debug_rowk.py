import os, sys, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
if len(sys.argv) > 1:
    import torch
    from fastvim_amd import mixer_ops as M
    torch.manual_seed(0)
    B, rows, cols, d_in = 2, 14, 14, 384
    dt = torch.bfloat16 if sys.argv[1] == "bf16" else torch.float32
    xz = torch.randn(B, rows * cols, 2 * d_in, device="cuda").to(dt)
    cw, cwb = torch.randn(d_in, 4, device="cuda") * .5, torch.randn(d_in, 4, device="cuda") * .5
    cb, cbb = torch.randn(d_in, device="cuda") * .1, torch.randn(d_in, device="cuda") * .1
    D, Db = torch.randn(d_in, device="cuda"), torch.randn(d_in, device="cuda")
    xc, skip = M.conv_pool_fwd(xz, cw, cb, cwb, cbb, rows, cols, False, 0, 1.0, D=D, D_b=Db)
    torch.save((xc.float().cpu(), skip.float().cpu()), sys.argv[2])
else:
    for dt in ("fp32", "bf16"):
        outs = {}
        for name, env in (("generic", {"FASTVIM_FWD_ROWK": "0"}), ("np1", {"FASTVIM_FWD_NP": "1"}), ("np3", {})):
            f = f"/tmp/dbg_{dt}_{name}.pt"
            subprocess.run([sys.executable, __file__, dt, f], env={**os.environ, **env}, check=True)
            import torch
            outs[name] = torch.load(f)
        for name in ("np1", "np3"):
            for i, what in enumerate(("xc", "skip")):
                a, b = outs["generic"][i], outs[name][i]
                print(dt, name, what, "maxdiff", (a - b).abs().max().item(), "nan", torch.isnan(b).sum().item())
