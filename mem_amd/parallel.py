"""Data-parallel gradient exchange for the flat-buffer engine (replaces the implicit DDP reducer
of /root/reference/mem/run_mem_pretraining.py:365-367 and its C4 parameter broadcast).

One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI).  The engine lays the
flat gradient buffer out in reverse-layer order, so each bucket (head, block L-1 ... block 0,
embedding) is final the moment backward leaves that layer: ``GradReducer`` is the engine's
``grad_hook`` and enqueues one asynchronous all-reduce per bucket right there.  RCCL runs them on
its own stream behind an event on the compute stream, i.e. overlapped with the rest of backward;
``finish()`` joins them before the gradient norm / AdamW.  ~30 MB fp32 per ViT-B block: large
enough for xGMI link bandwidth, small enough to pipeline 14 messages per step.
Mean semantics (SUM / world) == DDP.  Works on CPU tensors with gloo for the CPU tests.
"""
import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, flat_g, buckets, flat_p=None, group=None, coalesce_small=0, force=False):
        self.flat_g, self.buckets, self.group = flat_g, buckets, group
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

    def __call__(self, bucket_index):
        if not self.active:
            return
        _, b0, b1 = self.buckets[bucket_index]
        view = self.flat_g[b0:b1]
        if self.use_avg:
            h = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self.handles.append((h, None))
        else:
            h = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.handles.append((h, view))

    def finish(self):
        for h, view in self.handles:
            h.wait()
            if view is not None:
                view.div_(self.world)
        self.handles = []
