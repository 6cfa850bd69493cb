"""-m "not gpu": the N>1 data-parallel gradient exchange (parallel.GradReducer) with world_size 2
on the gloo backend: bucketed async all-reduce == mean of the per-rank gradients, parameters are
broadcast from rank 0, and the packed metric all-reduce of MetricLogger."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mem_amd.parallel import GradReducer
    from mem_amd import utils
    n = 5 * 1024
    buckets = [("head", 0, 1024), ("block1", 1024, 3072), ("block0", 3072, 4096), ("embed", 4096, n)]
    flat_p = torch.full((n,), float(rank + 1))
    flat_g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red = GradReducer(flat_g, buckets, flat_p=flat_p)
    assert torch.equal(flat_p, torch.ones(n))                    # rank-0 weights everywhere
    for b in range(len(buckets)):                                # the order backward releases buckets
        red(b)
    red.finish()
    want = torch.arange(n, dtype=torch.float32) * (1 + 2) / 2    # mean over ranks
    ok = torch.allclose(flat_g, want)
    ml = utils.MetricLogger()
    ml.update(loss=float(rank + 1), acc=0.5 * (rank + 1))
    ml.update(loss=float(rank + 3))
    ml.synchronize_between_processes()
    ok = ok and abs(ml.loss.global_avg - (1 + 3 + 2 + 4) / 4) < 1e-12 and ml.loss.count == 4
    ok = ok and abs(ml.acc.global_avg - (0.5 + 1.0) / 2) < 1e-12
    # the bad-sample flag is made global before it gates the update (NativeScalerWithGradNormCount): only rank 1 saw a bad
    # sample, both ranks hold the flag afterwards; a reducer that was released forgets its pending handles
    flag = torch.tensor([3 if rank == 1 else 0], dtype=torch.int64)
    red.sync_flag(flag)
    ok = ok and int(flag) == 3
    red(0)
    red.release()
    ok = ok and red.handles == []
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mem_amd.parallel import GradReducer
    n = 4 * 1024
    buckets = [("head", 0, 1024), ("block0", 1024, 3072), ("embed", 3072, n)]
    g = torch.Generator().manual_seed(100 + rank)
    flat_g = torch.randn(n, generator=g) * 0.01
    mine = flat_g.clone()
    others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) * 0.01 for r in range(world)]
    red = GradReducer(flat_g, buckets, bucket_dtype=torch.bfloat16)
    assert red.bytes_per_step == 2 * n
    for b in range(len(buckets)):
        red(b)
    red.finish()
    want32 = sum(others) / world                                  # the fp32 exchange
    # bf16_compress_hook arithmetic: every rank's bucket / world rounded to bf16, summed in bf16
    want16 = sum((o / world).bfloat16() for o in others).float()
    ok = torch.equal(flat_g, want16) and flat_g.dtype == torch.float32
    rel = float((flat_g - want32).norm() / want32.norm())
    ok = ok and rel < 8e-3 and torch.equal(mine, others[rank])
    q.put((rank, bool(ok), rel))
    dist.destroy_process_group()


def test_grad_reducer_bf16_buckets_world2_gloo():
    """The bf16-bucket option (184 MB instead of 367 MB per ViT-B step): two ranks end with IDENTICAL gradients, equal to
    the bf16 compress-hook arithmetic and within bf16 rounding (< 1 %) of the fp32 mean."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bf16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[:2] for r in res) == [(0, True), (1, True)], res


def test_reducer_single_process_noop():
    from mem_amd.parallel import GradReducer
    g = torch.ones(2048)
    r = GradReducer(g, [("a", 0, 1024), ("b", 1024, 2048)])
    r(0); r(1); r.finish()
    assert torch.equal(g, torch.ones(2048))


# ---------------------------------------------------------------------------------------------------------------------
# train_one_epoch on two gloo ranks with UNEQUAL masked-token counts: the reference's data-parallel semantics are a
# mean of per-rank means -- every rank's loss is the mean over ITS masked tokens (nn.CrossEntropyLoss() on
# outputs[bool_masked_pos], mem/engine_for_pretraining.py:152) and DDP averages the ranks' gradients with equal weights
# (mem/run_mem_pretraining.py:365-367), whatever the token counts.  The product's loop (engine_for_pretraining.train_one_epoch
# -> NativeScaler -> model.backward -> GradReducer buckets from the backward hook -> finish -> norm -> step) is run
# with a small CPU model that implements the engine interface the loop uses (the HIP engine needs a GPU).
class _ToyEngine:
    def __init__(self, n):
        self.flat_p = torch.zeros(n)
        self.flat_g = torch.zeros(n)
        self.buckets = [("head", 0, n // 2), ("embed", n // 2, n)]
        self.grad_hook = None
        self.gnorm = torch.zeros(1)

    def grad_norm(self):
        self.gnorm = self.flat_g.norm().reshape(1)
        return self.gnorm


class _ToyModel(torch.nn.Module):
    """logits[token] = x[token] @ W (D -> V), loss = mean CE over the rank's masked tokens; W lives in the flat buffers."""
    D, V = 8, 16

    def __init__(self):
        super().__init__()
        n = self.D * self.V
        self.engine = _ToyEngine(n)
        g = torch.Generator().manual_seed(7)
        self.engine.flat_p.copy_(torch.randn(n, generator=g) * 0.1)
        self.patch_embed = torch.nn.Module()
        self.patch_embed.proj = torch.nn.Conv2d(2, 4, 1)        # in_chans = 2 (the loop's 3 -> 2 channel view)
        self._fused_loss_pending = False

    def W(self):
        return self.engine.flat_p.view(self.D, self.V)

    def forward_loss(self, samples, bool_masked_pos, labels):
        x = samples.flatten(2).transpose(1, 2)[..., : self.D]   # [B, tokens, D]
        xm = x[bool_masked_pos]                                  # the rank's masked tokens
        W = self.W().clone().requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(xm @ W, labels)
        self._saved = (loss, W)
        acc = ((xm @ W).argmax(-1) == labels).float().mean()
        return torch.stack([loss.detach(), acc])

    def backward(self):
        loss, W = self._saved
        loss.backward()
        eng = self.engine
        eng.flat_g.copy_(W.grad.reshape(-1))
        for b in range(len(eng.buckets)):                        # buckets become final in this order
            if eng.grad_hook:
                eng.grad_hook(b)


class _ToyOpt:
    def __init__(self, model):
        self.engine = model.engine
        self.param_groups = [dict(lr=0.5, weight_decay=0.0, lr_scale=1.0)]
        self.max_norm = 0.0

    def step(self):
        self.engine.flat_p.sub_(self.param_groups[0]["lr"] * self.engine.flat_g)


class _ToyVae:
    def get_codebook_indices(self, images):
        return (images.flatten(2).sum(1) * 1000).long().abs() % _ToyModel.V      # [B, tokens]


def _toy_batch(rank):
    """(samples [B,8,4,4], images, mask [B,4,4]): rank 0 masks 3 tokens in all, rank 1 masks 11."""
    g = torch.Generator().manual_seed(50 + rank)
    samples = torch.randn(2, 8, 4, 4, generator=g)               # 8 "channels" = the D = 8 features of a token
    images = torch.randn(2, 3, 4, 4, generator=g)
    mask = torch.zeros(2, 16, dtype=torch.bool)
    idx = torch.randperm(32, generator=g)[: (3 if rank == 0 else 11)]
    mask.view(-1)[idx] = True
    return samples, images, mask.view(2, 4, 4)


def _toy_rank_grad(rank, W0):
    samples, images, mask = _toy_batch(rank)
    labels = _ToyVae().get_codebook_indices(images).flatten(1)[mask.flatten(1)]
    x = samples.flatten(2).transpose(1, 2)[..., : _ToyModel.D][mask.flatten(1)]
    W = W0.clone().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(x @ W, labels)
    loss.backward()
    return float(loss), W.grad.reshape(-1), int(mask.sum())


def _worker_loop(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mem_amd.parallel import GradReducer
    from mem_amd import utils
    from mem_amd.engine_for_pretraining import train_one_epoch
    model = _ToyModel()
    eng = model.engine
    W0 = model.W().clone()
    model._reducer = GradReducer(eng.flat_g, eng.buckets, flat_p=eng.flat_p)
    eng.grad_hook = model._reducer
    opt = _ToyOpt(model)
    samples, images, mask = _toy_batch(rank)
    loader = [((samples, images, mask), None)]
    stats = train_one_epoch(model, _ToyVae(), loader, opt, torch.device("cpu"), 0, utils.NativeScalerWithGradNormCount(),
                            max_norm=0, lr_schedule_values=[0.5], start_steps=0)
    # what the reference computes: every rank its own token mean, gradients averaged with equal weights
    l0, g0, n0 = _toy_rank_grad(0, W0)
    l1, g1, n1 = _toy_rank_grad(1, W0)
    assert n0 != n1
    want_g = (g0 + g1) / 2
    pooled_g = (g0 * n0 + g1 * n1) / (n0 + n1)                   # the global token mean: NOT what DDP computes
    ok = torch.allclose(eng.flat_g, want_g, atol=1e-6) and not torch.allclose(want_g, pooled_g, atol=1e-4)
    ok = ok and torch.allclose(eng.flat_p, W0.reshape(-1) - 0.5 * want_g, atol=1e-6)
    ok = ok and abs(stats["loss"] - (l0 + l1) / 2) < 1e-5        # MetricLogger.synchronize_between_processes: mean of means
    q.put((rank, bool(ok), float(eng.flat_p.sum())))
    dist.destroy_process_group()


def test_train_one_epoch_world2_mean_of_means():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_loop, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r[:2] for r in res) == [(0, True), (1, True)], res
    assert res[0][2] == res[1][2]                                # replicas hold identical parameters after the step


def _worker_dp_resume(rank, world, port, q, tmp):
    """the checkpoint holds ONE drop-path generator state PER RANK; after a resume every rank continues its own
    stream (replicas keep drawing different masks), and a checkpoint of another world size falls back to seed + rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    from pathlib import Path
    from mem_amd import utils

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))

    class Scaler:
        def state_dict(self): return {}
        def load_state_dict(self, s): pass

    def fresh(seed=7):
        m = Model()
        m._dp_stream = utils.DropPathStream()
        m._dp_stream.seed(seed + utils.get_rank())           # what run_mem_pretraining does before auto_load_model
        return m

    args = argparse.Namespace(output_dir=tmp, auto_resume=True, resume="", epochs=10, start_epoch=0)
    m = fresh()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    m._dp_stream.uniform(12, 8)                              # the run has drawn some masks before the checkpoint
    utils.save_model(args=args, epoch=0, model=m, model_without_ddp=m, optimizer=opt, loss_scaler=Scaler())
    cont = m._dp_stream.uniform(12, 8)                       # what THIS rank would have drawn next
    dist.barrier()
    ck = torch.load(Path(tmp) / "checkpoint-0.pth", map_location="cpu", weights_only=False)
    ok = ck["drop_path_rng"]["world"] == world and len(ck["drop_path_rng"]["states"]) == world
    ok = ok and not torch.equal(ck["drop_path_rng"]["states"][0], ck["drop_path_rng"]["states"][1])
    m2 = fresh(seed=1234)                                    # a resumed job (even one started with another seed)
    utils.auto_load_model(args=args, model=m2, model_without_ddp=m2, optimizer=torch.optim.SGD(m2.parameters(), lr=0.1),
                          loss_scaler=Scaler())
    mine = m2._dp_stream.uniform(12, 8)
    ok = ok and torch.equal(mine, cont)                      # every rank continues ITS stream
    both = [None, None]
    dist.all_gather_object(both, mine)
    ok = ok and not torch.equal(both[0], both[1])            # and the replicas' masks differ
    # a checkpoint written by a job of another world size: no state for this rank -> the stream keeps seed + rank
    m3 = fresh(seed=99)
    want = fresh(seed=99)._dp_stream.uniform(4, 4)
    loaded = utils.restore_rank_state(m3._dp_stream, {"world": 4, "states": ck["drop_path_rng"]["states"] * 2})
    ok = ok and not loaded and torch.equal(m3._dp_stream.uniform(4, 4), want)
    loaded = utils.restore_rank_state(m3._dp_stream, ck["drop_path_rng"]["states"][0])     # round-5 format in a 2-rank job
    ok = ok and not loaded
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_drop_path_stream_resumes_per_rank(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dp_resume, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]
