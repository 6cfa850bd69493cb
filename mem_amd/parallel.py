"""Data-parallel gradient exchange for the flat-buffer engine (replaces the implicit DDP reducer
of /root/reference/mem/run_mem_pretraining.py:365-367 and its C4 parameter broadcast).

One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI).  The engine lays the
flat gradient buffer out in reverse-layer order, so each bucket (head, block L-1 ... block 0,
embedding) is final the moment backward leaves that layer: ``GradReducer`` is the engine's
``grad_hook`` and enqueues one asynchronous all-reduce per bucket right there.  RCCL runs them on
its own stream behind an event on the CURRENT stream of the call -- the engine calls the hook with its
weight-gradient (side) stream current, after that stream has waited for the main stream's position
(``ViTEngine._bucket_ready``), so the collective is ordered behind every writer of the bucket while
the main stream never waits for a weight gradient -- i.e. overlapped with the rest of backward;
``finish()`` joins them before the gradient norm / AdamW.  ~30 MB fp32 per ViT-B block: large
enough for xGMI link bandwidth, small enough to pipeline 14 messages per step.
Mean semantics (SUM / world) == DDP.  Works on CPU tensors with gloo for the CPU tests.

Options (both off by default = the reference's fp32 DDP exchange):
  * ``bucket_dtype=torch.bfloat16``: a bucket is rounded to bf16, pre-divided by the world size, summed over the ranks in
    bf16 and widened back (torch's bf16_compress_hook arithmetic): 184 MB instead of 367 MB per ViT-B step on the wire;
  * ``reserve_cus=k``: while buckets are in flight the library sizes the persistent one-workgroup-per-CU grids (GEMMs,
    attention) it launches ON THIS ENGINE'S STREAMS for k CUs fewer (``memhip_stream_reserve_cus``: scoped to the stream
    handles, another engine of the process is not affected), so that RCCL's channel kernels find CUs of their own instead
    of displacing workgroups of a grid that covers the whole chip (whose stragglers then run a second round).  The
    reservation is dropped by ``finish()`` / ``release()``; the training loops call ``release()`` on every exit path.
    No measurement on more than one GPU exists yet: the entrypoint leaves it at 0.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat_g, buckets, flat_p=None, group=None, coalesce_small=0, force=False, bucket_dtype=None,
                 reserve_cus=0, streams=None):
        """streams: callable returning the torch.cuda.Stream objects the engine launches on (main + side stream): the CU
        reservation is set on exactly these."""
        self.flat_g, self.buckets, self.group = flat_g, buckets, group
        self._streams_fn, self._res_streams = streams, []
        self.bucket_dtype = bucket_dtype if bucket_dtype not in (None, torch.float32) else None
        self.reserve_cus = int(reserve_cus)
        self._reserved = False
        self.wire = torch.empty_like(flat_g, dtype=self.bucket_dtype) if self.bucket_dtype is not None else None
        esz = 2 if self.bucket_dtype is not None else 4
        self.bytes_per_step = sum(b1 - b0 for _, b0, b1 in buckets) * esz
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force: issue the collectives even in a one-rank group (single-GPU dry run of the RCCL path: bench.py
        # with MEMHIP_BENCH_FORCE_DIST=1)
        self.active = (self.world > 1 or force) and dist.is_initialized()
        self.handles = []
        # RCCL (ncclAvg, RCCL >= 2.10) averages inside the collective; gloo (CPU tests) has no AVG: SUM, then divide
        # on the compute stream after the join (one extra pass over the buckets)
        self.use_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        if flat_p is not None and self.active:
            dist.broadcast(flat_p, src=0, group=group)          # rank-0 weights everywhere (DDP ctor)

    def _reserve(self, on):
        if self.reserve_cus > 0 and self._reserved != on and self.flat_g.is_cuda:
            from . import ops
            if on:
                self._res_streams = [s for s in (self._streams_fn() if self._streams_fn else [torch.cuda.current_stream()])
                                     if s is not None]
            for s in self._res_streams:
                ops.stream_reserve_cus(s, self.reserve_cus if on else 0)
            self._reserved = on

    def release(self):
        """Drop the CU reservation and forget pending handles (exception / early-return paths of a training loop)."""
        self._reserve(False)
        self.handles = []

    def __del__(self):
        try:
            self._reserve(False)
        except Exception:
            pass

    def __call__(self, bucket_index):
        if not self.active:
            return
        self._reserve(True)                       # from the first bucket of a step until finish()
        _, b0, b1 = self.buckets[bucket_index]
        view = self.flat_g[b0:b1]
        if self.wire is not None:
            w = self.wire[b0:b1]
            # pre-divide in fp32, ONE rounding to bf16 (the division writes straight into the preallocated wire buffer: no
            # fp32 temporary of the bucket's size per call), then a bf16 SUM over the ranks
            torch.div(view, self.world, out=w)
            h = dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append((h, (view, w)))
            return
        if self.use_avg:
            h = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self.handles.append((h, None))
        else:
            h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append((h, view))

    def sync_flag(self, flag):
        """MAX of a small integer / float flag tensor over the ranks, in place (the bad-sample counter of the training loop:
        every rank must take the SAME skip-this-update decision and every rank must raise at the next check -- a rank-local
        flag would let the replicas diverge).  No-op outside a multi-rank group."""
        if self.active and self.world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        return flag

    def finish(self):
        for h, view in self.handles:
            h.wait()
            if isinstance(view, tuple):            # bf16 wire buffer -> the fp32 gradient
                view[0].copy_(view[1])
            elif view is not None:
                view.div_(self.world)
        self.handles = []
        self._reserve(False)
