"""The data-parallel training step as a chain of HIP graphs with the gradient exchange between them.

torch DDP (what the reference trains with, imagenet_classification/train.py:34-43) overlaps its bucketed all-reduce
with backward through autograd hooks.  Here the step is captured into HIP graphs, so the overlap is made explicit
(SURVEY.md section 8e): the block stack is cut into K runs of blocks; ONE graph holds the forward pass, K graphs hold
the backward pass of one run each (last run first), one graph holds the fused optimizer.  The flat gradient is laid
out block by block (fastvim_amd/flat.py), so the gradient of run k is one contiguous bucket: as soon as backward graph
k has been enqueued, bucket k's all-reduce is launched asynchronously (RCCL, on the process group's stream, ordered
behind graph k by an event) and runs under backward graphs k+1 .. K-1:

    replay(fwd) -> replay(bwd 0), launch(bucket 0) -> replay(bwd 1), launch(bucket 1) -> ... -> finish() -> replay(opt)

Only the last bucket (the first blocks + patch embedding) is exposed.  Cutting backward needs the activations at the
cuts to be graph leaves: each run's input (hidden, residual) is a detached view of the previous run's output, and
backward of run k is ``torch.autograd.backward(outputs_k, grads of run k+1's inputs)``.  With one rank nothing is
exchanged and the chain computes what the single-graph step computes: bit for bit with ``gemm.FILL = False``; with
the shipped default (``gemm.fill_splits`` re-cuts a segment's smaller weight-gradient groups into more K slices) the
same sums are taken in another order and the two agree to rounding (tests/test_pipeline_gpu.py asserts both).
"""
import warnings

import torch
import torch.distributed as dist


class SegmentedTrainStep:
    """``model`` must expose ``_embed / _run_layers / _final / _head`` (fastvim_amd.fastvim.VisionMamba).
    ``loss_fn(logits, target) -> scalar``.  ``x`` / ``target`` are the static input buffers the graphs read; copy
    new batches into them between steps.

    Capture needs ``warmup`` eager steps first (allocator, lazily built buffers, RCCL communicators).  They are real
    steps on whatever ``x`` / ``target`` hold, so the training state they touch -- parameters, bf16 shadow, Adam moments,
    EMA, step count, the CPU and GPU RNG streams -- is snapshotted before and put back after them: constructing the
    step object leaves the model, the optimizer and the DropPath stream exactly as it found them."""

    def __init__(self, model, flat, opt, loss_fn, x, target, n_segments=3, amp_dtype=torch.bfloat16, use_graph=True,
                 warmup=2):
        self.model, self.flat, self.opt, self.loss_fn = model, flat, opt, loss_fn
        self.x, self.target, self.amp_dtype = x, target, amp_dtype
        self.exchange = flat.make_exchange(n_segments)
        self.runs = self.exchange.layers                 # (lo, hi) block ranges in BACKWARD order
        self.K = len(self.runs)
        self._gscale = 1.0 / self.exchange.world_size
        self.use_graph = use_graph
        self.loss = None
        self._seed = torch.ones((), device=x.device, dtype=torch.float32)      # d loss / d loss, allocated once outside the graphs
        self._cuts = None
        self._exposed = []                               # (event before finish, event after) of the timed steps
        self.graphs = None
        if use_graph:
            self._capture(warmup)

    # ------------------------------------------------------------------ the pieces
    def _forward(self):
        m = self.model
        self.flat.zero_grad()
        cuts = []              # per run in FORWARD order: (inputs (leaves) or None, outputs)
        with torch.autocast("cuda", dtype=self.amp_dtype, enabled=self.amp_dtype != torch.float32):
            h, _ = m._embed(self.x)
            res, pend = None, None
            fwd_runs = self.runs[::-1]
            open_runs = hasattr(m, "_run_layers_open")
            for i, (lo, hi) in enumerate(fwd_runs):
                if i > 0:          # cut: this run's inputs are leaves that alias the previous run's outputs
                    res = res.detach().requires_grad_()
                    if pend is not None:      # the chain of fused blocks continues across the cut (fastvim._run_layers_open):
                        pend = (pend[0].detach().requires_grad_(), pend[1])      # its hand-over is the gated activations
                        ins = (pend[0], res)
                    else:
                        h = h.detach().requires_grad_()
                        ins = (h, res)
                else:
                    ins = None
                if open_runs:
                    h, res, pend = m._run_layers_open(h, res, lo, hi, pend=pend)
                else:
                    h, res = m._run_layers(h, res, lo, hi)
                cuts.append([ins, (h if pend is None else pend[0], res)])
            if pend is not None:
                h = m._close_run(pend)
            logits = m._head(m._final(h, res))
        loss = self.loss_fn(logits, self.target)
        cuts[-1][1] = (loss,)
        self._cuts = cuts
        return loss.detach()

    def _backward(self, k):
        """Backward of run k (backward order: k = 0 is the last run of blocks + head + loss)."""
        i = self.K - 1 - k                      # forward index
        outs = self._cuts[i][1]
        if k == 0:
            seed = self._seed if (outs[0].dim() == 0 and outs[0].dtype == self._seed.dtype and outs[0].device == self._seed.device) else None
            torch.autograd.backward(outs[0], seed)      # (the seed autograd would otherwise fill inside the captured graph)
        else:
            nxt = self._cuts[i + 1][0]
            torch.autograd.backward(list(outs), [t.grad for t in nxt])
        self.flat.finish_backward()             # this run's queued weight-gradient GEMMs and partial sums
        self._cuts[i][1] = None                 # free the run's autograd graph

    # ------------------------------------------------------------------ capture / run
    def _eager_step(self):
        loss = self._forward()
        for k in range(self.K):
            self._backward(k)
            self.exchange.launch(k)
        self.exchange.finish(mean=False)         # sums: the optimizer kernel applies 1 / world as it reads the gradient
        self.opt.step(grad_scale=self._gscale)
        return loss

    def _snapshot(self):
        f, o = self.flat, self.opt
        # every tensor attribute of the optimizer (moments, EMA, step count, lr, decay mask, ...) and the state's own buffers:
        # whatever a warm-up step may advance is put back, not a hand-picked list
        bufs = [f.param_flat, f.shadow_flat] + list(getattr(f, "_t_dst", [])) + list(getattr(f, "_tx_dst", []))   # (+ the transposed in_proj / x_proj shadows)
        for v in vars(o).values():
            if torch.is_tensor(v) and v.is_cuda and all(v is not b for b in bufs):
                bufs.append(v)
        dev = f.param_flat.device
        return ([(b, b.clone()) for b in bufs], torch.get_rng_state(), torch.cuda.get_rng_state(dev), dev)

    @staticmethod
    def _restore(snap):
        bufs, cpu_rng, gpu_rng, dev = snap
        with torch.no_grad():
            for b, saved in bufs:
                b.copy_(saved)
        torch.set_rng_state(cpu_rng)
        torch.cuda.set_rng_state(gpu_rng, dev)

    def _capture(self, warmup):
        from . import graph_capture_safe
        if not graph_capture_safe():
            # known to replay some captured steps wrongly (non-finite after a few replays): do not capture at all
            warnings.warn("SegmentedTrainStep: HIP was initialised before `import fastvim_amd` could switch graph packet "
                          "capture off (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0): on ROCm 7.2 some captured training steps replay "
                          "wrongly (DESIGN.md section 5) -- running this step EAGERLY instead.  Import fastvim_amd first, or "
                          "export the variable, to get graph replay.", RuntimeWarning, stacklevel=3)
            self.use_graph = False
            return
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._restore(snap)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        # With a process group alive, its watchdog THREAD polls the events of finished collectives (hipEventQuery); under the
        # default "global" capture mode any thread's event query while this thread captures is an error
        # (hipErrorStreamCaptureUnsupported: "operation not permitted when stream is capturing" -- it took the whole process
        # down, intermittently, in tests/test_pipeline_gpu.py::test_rccl_buckets_between_backward_graphs).  "thread_local"
        # restricts the check to the capturing thread, which issues nothing but kernel launches.
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fwd, pool=pool, capture_error_mode=mode):
            self.loss = self._forward()
        g_bwd = []
        for k in range(self.K):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode=mode):
                self._backward(k)
            g_bwd.append(g)
        g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_opt, pool=pool, capture_error_mode=mode):
            self.opt.step(grad_scale=self._gscale)
        self.graphs = (g_fwd, g_bwd, g_opt)

    def step(self, time_exposed=False):
        """One training step; returns the (device) loss tensor.  ``time_exposed`` brackets the wait for the gradient
        exchange with events (read them with ``exposed_ms()`` after a synchronize)."""
        if not self.use_graph:
            self.loss = self._eager_step()
            return self.loss
        g_fwd, g_bwd, g_opt = self.graphs
        g_fwd.replay()
        for k in range(self.K):
            g_bwd[k].replay()
            self.exchange.launch(k)
        if time_exposed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.exchange.finish(mean=False)
        if time_exposed:
            e1.record()
            self._exposed.append((e0, e1))
        g_opt.replay()
        return self.loss

    def exposed_ms(self):
        """Mean time the compute stream waited for the gradient exchange after the last backward graph (call after a
        device synchronize)."""
        if not self._exposed:
            return None
        t = [a.elapsed_time(b) for a, b in self._exposed]
        self._exposed = []
        return sum(t) / len(t)
