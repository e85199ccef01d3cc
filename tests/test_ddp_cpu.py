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
