"""world_size-2 gloo test of the data-parallel gradient exchange (fastvim_amd.ddp): flat-buffer
all-reduce == gradient of the full batch on one process; chunking and bf16 compression paths."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 4))


def _worker(rank, world, port, chunk_bytes, comm_dtype, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastvim_amd.ddp import FlatGradAllReduce, shard_batch
    m = _model()
    flat = FlatGradAllReduce(m.parameters(), chunk_bytes=chunk_bytes, comm_dtype=comm_dtype)
    x = torch.randn(8, 16, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 4, generator=torch.Generator().manual_seed(2))
    lo, hi = shard_batch(8, rank, world)
    for _ in range(2):                       # second iteration checks zero_() + in-place accumulation into the views
        flat.zero_()
        ((m(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
        flat.allreduce_mean_()
    assert all(p.grad.data_ptr() >= flat.flat.data_ptr() for p in m.parameters())
    if rank == 0:
        torch.save(flat.flat.clone(), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chunk_bytes,comm_dtype,tol", [(256 << 20, None, 1e-6), (64, None, 1e-6),
                                                         (256 << 20, torch.bfloat16, 2e-2)])
def test_flat_grad_allreduce_matches_full_batch(tmp_path, chunk_bytes, comm_dtype, tol):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), chunk_bytes, comm_dtype, out), nprocs=2, join=True)
    got = torch.load(out)
    m = _model()
    x = torch.randn(8, 16, generator=torch.Generator().manual_seed(1))
    y = torch.randn(8, 4, generator=torch.Generator().manual_seed(2))
    ((m(x) - y) ** 2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    assert (got - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())


def _bucket_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastvim_amd.ddp import GradExchange
    g = torch.Generator().manual_seed(10 + rank)
    base = torch.randn(1000, generator=g)
    res = {}
    for name, bounds, order in (("one", None, [0]), ("four", [(700, 1000), (400, 700), (96, 400), (0, 96)], [0, 1, 2, 3]),
                                ("chunked", [(512, 1000), (0, 512)], [0, 1])):
        flat = base.clone()
        ex = GradExchange(flat, bounds, chunk_bytes=(1 << 30) if name != "chunked" else 400)
        for k in order:             # buckets launched one by one (as backward segments complete), finished once
            ex.launch(k)
        ex.finish()
        res[name] = flat
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_exchange_bitwise_equals_single_allreduce(tmp_path):
    """The bucketed, asynchronously launched exchange computes exactly the single all-reduce's mean."""
    out = str(tmp_path / "b.pt")
    mp.spawn(_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    ref = (torch.randn(1000, generator=torch.Generator().manual_seed(10))
           + torch.randn(1000, generator=torch.Generator().manual_seed(11))) / 2
    assert torch.equal(res["one"], ref)
    assert torch.equal(res["four"], res["one"]) and torch.equal(res["chunked"], res["one"])


def _auto_wire_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastvim_amd.ddp import GradExchange
    g = torch.Generator().manual_seed(20 + rank)
    base = torch.randn(4096, generator=g) * torch.logspace(-6, 2, 4096)      # eight decades of gradient magnitude
    bounds = [(1024, 4096), (0, 1024)]
    res = {}
    # "auto" with the threshold between the two bucket sizes: the 12 KiB bucket goes over the wire as bf16, the 4 KiB one as fp32
    ex = GradExchange(base.clone(), bounds, comm_dtype="auto", bf16_min_bytes=8 << 10)
    res["wire"] = ex.wire_names()
    ex.allreduce_(mean=False)
    res["auto_sum"] = ex.flat
    ex32 = GradExchange(base.clone(), bounds, comm_dtype=None)
    ex32.allreduce_(mean=False)
    res["fp32_sum"] = ex32.flat
    exm = GradExchange(base.clone(), bounds, comm_dtype=None)
    exm.allreduce_mean_()
    res["fp32_mean"] = exm.flat
    res["default_T"] = GradExchange(torch.zeros(7_200_000), None, comm_dtype="auto").wire_names()       # 28.8 MB: fp32
    res["default_B"] = GradExchange(torch.zeros(40_000_000), None, comm_dtype="auto").wire_names()      # 160 MB: bf16
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_auto_wire_format_keeps_fp32_accumulation_error_bound(tmp_path):
    """``comm_dtype="auto"``: buckets of >= bf16_min_bytes travel as bf16, smaller ones as fp32.  The gradient stays fp32
    on both sides of the wire, so the error of a bf16 bucket is the rounding of each addend and of the wire sum: at most
    2^-8 of (|a| + |b|) + 2^-8 of |a + b| per element -- checked elementwise over eight decades of magnitude; the fp32
    bucket is exact.  Sums followed by the optimizer's ``grad_scale = 1 / world`` equal the mean for a power-of-two world."""
    out = str(tmp_path / "w.pt")
    mp.spawn(_auto_wire_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["wire"] == ["bfloat16", "float32"] and r["default_T"] == ["float32"] and r["default_B"] == ["bfloat16"]
    sc = torch.logspace(-6, 2, 4096)
    a = torch.randn(4096, generator=torch.Generator().manual_seed(20)) * sc
    b = torch.randn(4096, generator=torch.Generator().manual_seed(21)) * sc
    assert torch.equal(r["fp32_sum"], a + b)
    assert torch.equal(r["auto_sum"][:1024], (a + b)[:1024])                        # the small bucket went as fp32
    err = (r["auto_sum"][1024:] - (a + b)[1024:]).abs()
    bound = 2.0 ** -8 * (a.abs() + b.abs() + (a + b).abs())[1024:]
    assert (err <= bound).all() and err.max() > 0                                   # bf16 on the wire, within its bound
    assert r["auto_sum"].dtype == torch.float32
    assert torch.equal(r["fp32_sum"] * 0.5, r["fp32_mean"])                         # grad_scale = 1 / 2 == the mean


def _ring8_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fastvim_amd.ddp import GradExchange
    base = torch.randn(4096, generator=torch.Generator().manual_seed(40 + rank)) * torch.logspace(-6, 2, 4096)
    ex = GradExchange(base.clone(), None, comm_dtype=torch.bfloat16)
    ex.allreduce_(mean=False)
    ex32 = GradExchange(base.clone(), None)              # the default wire: fp32
    ex32.allreduce_(mean=False)
    if rank == 0:
        torch.save({"bf16": ex.flat, "fp32": ex32.flat, "wire32": ex32.wire_names()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_wire_error_bound_with_eight_ranks(tmp_path):
    """The opt-in bf16 wire at the node's full width: eight addends, each rounded once to bf16, summed IN bf16 (one more
    rounding per partial sum, whatever order the backend takes them in) -- |error| <= 2^-8 * (sum |a_i| + 7 * sum |a_i|) per
    element, and in practice a relative error of a few 2^-9 of sum |a_i|.  The default wire (fp32) differs from the fp64
    sum only by fp32 rounding.  This is why bf16 is not the default: the reference's DDP sums fp32
    (imagenet_classification/train.py:34-43)."""
    import inspect
    from fastvim_amd.flat import FlatTrainingState
    assert inspect.signature(FlatTrainingState.__init__).parameters["comm_dtype"].default is None
    out = str(tmp_path / "r8.pt")
    mp.spawn(_ring8_worker, args=(8, _free_port(), out), nprocs=8, join=True)
    r = torch.load(out)
    sc = torch.logspace(-6, 2, 4096)
    parts = [torch.randn(4096, generator=torch.Generator().manual_seed(40 + k)) * sc for k in range(8)]
    exact = torch.stack(parts).double().sum(0)
    mag = torch.stack(parts).double().abs().sum(0)
    assert r["wire32"] == ["float32"]
    assert ((r["fp32"].double() - exact).abs() <= 8 * 2.0 ** -24 * mag).all()
    err = (r["bf16"].double() - exact).abs()
    assert (err <= 8 * 2.0 ** -8 * mag).all() and err.max() > 0
    assert (err / mag).mean() < 2.0 ** -8          # typical error: well inside the worst case


def test_layer_major_buckets_tile_the_flat_gradient():
    """FlatTrainingState lays the gradient out block by block; buckets() cuts it into runs of whole blocks, last
    blocks first, that tile the buffer (CPU: no kernels involved)."""
    from fastvim_amd.fastvim import VisionMamba
    from fastvim_amd.flat import FlatTrainingState
    m = VisionMamba(img_size=32, depth=7, embed_dim=32, num_classes=5, rms_norm=True, fused_add_norm=True, residual_in_fp32=True)
    with FlatTrainingState(m) as flat:
        named = dict(m.named_parameters())
        for n in (1, 2, 3, 4, 7, 12):
            bk = flat.buckets(n)
            assert len(bk) == min(n, 7)
            assert bk[0]["bounds"][1] == flat.grad_flat.numel() and bk[-1]["bounds"][0] == 0
            assert all(a["bounds"][0] == b["bounds"][1] for a, b in zip(bk[:-1], bk[1:]))
            assert bk[0]["layers"][1] == 7 and bk[-1]["layers"][0] == 0
            for b in bk:                                   # every block's parameters sit inside its bucket
                for i in range(*b["layers"]):
                    for pn, p in named.items():
                        if pn.startswith(f"layers.{i}."):
                            assert b["bounds"][0] <= flat.offsets[pn] and flat.offsets[pn] + p.numel() <= b["bounds"][1]
            assert bk[0]["bounds"][0] <= flat.offsets["norm_f.weight"]                 # after the stack: first bucket
            assert flat.offsets["patch_embed.proj.weight"] < bk[-1]["bounds"][1]       # before it: last bucket
