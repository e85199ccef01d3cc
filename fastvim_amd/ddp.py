"""Data-parallel gradient exchange for the FastVim training step.

The reference has no communication code of its own: Lightning wraps the model in torch DDP over NCCL
(imagenet_classification/train.py:34-43), which all-reduces 25 MB buckets in reverse layer order while
backward is still running.  Here it is explicit: one process per GPU, gradients live in ONE flat fp32
buffer laid out layer by layer (``p.grad`` are views, fastvim_amd/flat.py), and ``GradExchange`` sums
contiguous BUCKETS of that buffer across ranks with RCCL (``torch.distributed`` backend "nccl" on ROCm):

* ``launch(k)`` enqueues the all-reduce of bucket k asynchronously: the collective waits (by event) for the
  work already queued on the current stream -- the backward segment that produced the bucket -- and runs on the
  process group's own stream, so the next backward segment overlaps it;
* ``finish()`` makes the current stream wait for every outstanding bucket and divides by the world size
  (torch DDP's mean semantics) -- the optimizer goes after it.  ``finish(mean=False)`` leaves the SUMS: the fused
  optimizer kernel takes ``1 / world`` as its gradient scale (``FlatAdamW.step(grad_scale=...)``), which saves a full
  read-modify-write pass over the gradient per step (391 MB at FastVim-B).

xGMI is point-to-point (7 links per GPU), so ring collectives are per-link bound: a few LARGE buckets (default:
3 per step, tapered 10 : 8 : 6 blocks -- 12 / 9 / 7 MB at FastVim-T, 160 / 130 / 100 MB at FastVim-B -- so that the
one nothing overlaps is the smallest; fastvim_amd/flat.py ``buckets``) instead of torch DDP's 25 MB default.  The wire
format is fp32 by default -- what the reference's DDP sums.  A bf16 wire that halves the bytes per link is an explicit
opt-in (``comm_dtype=torch.bfloat16``, or ``"auto"``: bf16 for buckets whose fp32 size is at least ``bf16_min_bytes`` =
100 MB, i.e. the FastVim-B buckets where the exchange is bandwidth-bound; ``bench.py --comm-dtype``), like the reference's
detection-only fp16 hook (detection/vitdet/fp16_compression_hook.py:17-26).  With it the gradient stays fp32 on both sides
of the wire: a bucket is rounded once to bf16, summed by RCCL IN bf16 (a ring of N ranks rounds once per hop: the error
grows with N, tests/test_ddp_cpu.py bounds N = 2 and N = 8), and written back into the fp32 buffer.
No BatchNorm exists in FastVim, so nothing else is
exchanged.  The same class is the whole-buffer exchange (one bucket) used by
``FlatTrainingState.allreduce_mean_`` and by the CPU (gloo) tests.
"""
import torch
import torch.distributed as dist


class GradExchange:
    """All-reduce (mean) of contiguous buckets of one flat fp32 gradient buffer.

    ``bounds``: list of (lo, hi) element ranges, in the order the backward pass completes them; default one bucket
    covering the buffer.  ``chunk_bytes`` caps the size of a single collective call (very large buckets are sent
    as several calls).  ``comm_dtype``: wire dtype -- None / torch.float32, torch.bfloat16, or "auto" (bf16 for buckets of
    at least ``bf16_min_bytes`` of fp32 gradient, fp32 below)."""

    def __init__(self, flat_grad, bounds=None, process_group=None, comm_dtype=None, chunk_bytes=256 << 20,
                 bf16_min_bytes=100 << 20):
        assert flat_grad.dtype == torch.float32 and flat_grad.is_contiguous() and flat_grad.dim() == 1
        self.flat = flat_grad
        self.group = process_group
        self.chunk = max(1, chunk_bytes // 4)
        self.bounds = [(0, flat_grad.numel())] if bounds is None else [(int(a), int(b)) for a, b in bounds]
        if isinstance(comm_dtype, str):
            assert comm_dtype == "auto", comm_dtype
            self.wire_dtypes = [torch.bfloat16 if (b - a) * 4 >= bf16_min_bytes else None for a, b in self.bounds]
        else:
            self.wire_dtypes = [None if comm_dtype in (None, torch.float32) else comm_dtype] * len(self.bounds)
        self.comm_dtype = comm_dtype
        covered = sorted(self.bounds)
        assert covered[0][0] == 0 and covered[-1][1] == flat_grad.numel() and \
            all(a[1] == b[0] for a, b in zip(covered[:-1], covered[1:])), "buckets must tile the gradient buffer"
        self._pending = []          # (work handle, view, wire buffer or None)
        self._wire = {}             # persistent wire buffers (graph-safe addresses, no per-step allocation)
        # per-bucket timing (bench.py --gpus N): events bracketing each bucket's collective ON A STREAM OF ITS OWN -- the
        # first waits for the producer (the backward segment), the second for the collective's completion -- so the
        # duration is launch-to-completion of the exchange itself, whatever the compute stream does meanwhile
        self.timing = False
        self._cs = None
        self._timed = []            # (bucket, start event, end event)

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def launch(self, k):
        """Start summing bucket k across ranks (asynchronous; returns immediately)."""
        if self.world_size == 1:
            return
        lo, hi = self.bounds[k]
        if self.timing and self.flat.is_cuda:
            return self._launch_timed(k)
        for s in range(lo, hi, self.chunk):
            e = min(hi, s + self.chunk)
            view = self.flat[s:e]
            wire = self.wire_dtypes[k]
            if wire is not None:
                buf = self._wire.get((s, e))
                if buf is None:
                    buf = self._wire[(s, e)] = torch.empty(e - s, device=view.device, dtype=wire)
                buf.copy_(view)
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._pending.append((work, view, buf))
            else:
                work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._pending.append((work, view, None))

    def _launch_timed(self, k):
        """``launch`` with the bucket's collective bracketed by events on a side stream (see ``timing``)."""
        if self._cs is None:
            self._cs = torch.cuda.Stream(self.flat.device)
        cur = torch.cuda.current_stream(self.flat.device)
        self._cs.wait_stream(cur)
        lo, hi = self.bounds[k]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self._cs):
            e0.record()
            for s in range(lo, hi, self.chunk):
                e = min(hi, s + self.chunk)
                view = self.flat[s:e]
                wire = self.wire_dtypes[k]
                buf = None
                if wire is not None:
                    buf = self._wire.get((s, e))
                    if buf is None:
                        buf = self._wire[(s, e)] = torch.empty(e - s, device=view.device, dtype=wire)
                    buf.copy_(view)
                work = dist.all_reduce(buf if buf is not None else view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                work.wait()                      # (stream-wise: the side stream waits for the collective)
                if buf is not None:
                    view.copy_(buf)
            e1.record()
        self._timed.append((k, e0, e1))
        self._pending.append((None, None, None))      # finish() joins the side stream

    def bucket_ms(self):
        """Per bucket: mean launch-to-completion time of its collective over the timed steps (after a synchronize)."""
        out = {}
        for k, e0, e1 in self._timed:
            out.setdefault(k, []).append(e0.elapsed_time(e1))
        self._timed = []
        return [round(sum(out[k]) / len(out[k]), 3) if k in out else None for k in range(len(self.bounds))]

    def finish(self, mean=True):
        """Wait (stream-wise on GPUs) for every launched bucket; ``mean``: turn the sums into means (one more pass over
        the buffer -- pass False when the optimizer applies ``1 / world_size`` itself)."""
        ws = self.world_size
        if ws == 1:
            return
        pending, self._pending = self._pending, []
        for work, view, buf in pending:
            if work is None:                     # timed launch: everything happened on the side stream
                torch.cuda.current_stream(self.flat.device).wait_stream(self._cs)
                continue
            work.wait()
            if buf is not None:
                view.copy_(buf)
        if mean:
            self.flat.div_(ws)

    def wire_names(self):
        return [str(d or torch.float32).replace("torch.", "") for d in self.wire_dtypes]

    def allreduce_(self, mean=True):
        """Whole buffer in one go: every bucket launched, then finished."""
        for k in range(len(self.bounds)):
            self.launch(k)
        self.finish(mean=mean)

    def allreduce_mean_(self):
        self.allreduce_(mean=True)


class FlatGradAllReduce:
    """Flat gradient buffer for an arbitrary parameter list (``p.grad`` become views of ``flat``) plus its mean
    all-reduce: the stand-alone form of what ``FlatTrainingState`` does for a whole model."""

    def __init__(self, params, process_group=None, chunk_bytes=256 << 20, comm_dtype=None, n_buckets=1):
        self.params = [p for p in params if p.requires_grad]
        numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(numel, device=dev, dtype=torch.float32)
        off, offs = 0, []
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)   # autograd accumulates in place into the view
            offs.append(off)
            off += n
        # buckets of whole parameters, last parameters first (the order backward produces them)
        bounds = None
        if n_buckets > 1:
            cuts = sorted({offs[min(len(offs) - 1, round(i * len(offs) / n_buckets))] for i in range(1, n_buckets)} - {0})
            edges = [0] + cuts + [numel]
            bounds = list(reversed(list(zip(edges[:-1], edges[1:]))))
        self.exchange = GradExchange(self.flat, bounds, process_group, comm_dtype, chunk_bytes)

    def zero_(self):
        self.flat.zero_()

    @property
    def world_size(self):
        return self.exchange.world_size

    def allreduce_mean_(self):
        """Sum the flat gradient across ranks and divide by the world size (torch DDP semantics)."""
        self.exchange.allreduce_mean_()


def shard_batch(global_batch, rank, world_size):
    """Even split of a global batch; BASELINE configs give the per-GPU batch directly (weak scaling)."""
    assert global_batch % world_size == 0
    per = global_batch // world_size
    return rank * per, (rank + 1) * per
