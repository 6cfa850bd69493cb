"""Runtime utilities -- mirror of the pretraining-path parts of /root/reference/mem/utils.py:
SmoothedValue / MetricLogger (:34-183), distributed init (:235-299), NativeScalerWithGradNormCount
(:351-377), get_grad_norm_ (:380-392), cosine_scheduler (:395-412), save_model / auto_load_model
(:425-447, :485-519), create_d_vae / get_event_vae (:559-578).

SmoothedValue, MetricLogger, setup_for_distributed and cosine_scheduler are plain RESTATEMENTS of the reference's logging /
schedule helpers (DeiT-lineage boilerplate that is part of the named surface: meter names, formats and the returned dict are
observable), not designs of this repo; nothing on the hot path lives in them.

Differences that are deliberate: device-agnostic (the reference hard-codes
'cuda' / torch.cuda.synchronize), one packed all-reduce for all meters instead of one per meter,
and a scaler that is a no-op for bf16 (no loss scaling needed) but keeps the checkpoint key.
"""
import datetime
import glob
import math
import os
import time
from collections import defaultdict, deque
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist


class SmoothedValue(object):
    def __init__(self, window_size=20, fmt=None):
        if fmt is None:
            fmt = "{median:.4f} ({global_avg:.4f})"
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self.fmt = fmt

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """utils.py:52-63 (count/total only; the deque stays local)."""
        if not is_dist_avail_and_initialized():
            return
        t = _sync_device_tensor([self.count, self.total])
        self.count, self.total = int(t[0]), t[1]

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return float("nan") if self.count == 0 else self.total / self.count

    @property
    def max(self):
        return float("nan") if len(self.deque) == 0 else max(self.deque)

    @property
    def value(self):
        return float("nan") if len(self.deque) == 0 else self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


def _sync_device_tensor(values):
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(values, dtype=torch.float64, device=dev)
    dist.barrier()
    dist.all_reduce(t)
    return t.tolist()


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {}".format(n, str(m)) for n, m in self.meters.items())

    def synchronize_between_processes(self):
        """All meters in ONE packed fp64 all-reduce (the reference issues one per meter)."""
        if not is_dist_avail_and_initialized():
            return
        names = list(self.meters.keys())
        flat = []
        for n in names:
            flat += [self.meters[n].count, self.meters[n].total]
        t = _sync_device_tensor(flat)
        for i, n in enumerate(names):
            self.meters[n].count, self.meters[n].total = int(t[2 * i]), t[2 * i + 1]

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        i = 0
        header = header or ""
        start_time = time.time()
        end = time.time()
        iter_time = SmoothedValue(fmt="{avg:.4f}")
        data_time = SmoothedValue(fmt="{avg:.4f}")
        space_fmt = ":" + str(len(str(len(iterable)))) + "d"
        log_msg = [header, "[{0" + space_fmt + "}/{1}]", "eta: {eta}", "{meters}", "time: {time}", "data: {data}"]
        if torch.cuda.is_available():
            log_msg.append("max mem: {memory:.0f}")
        log_msg = self.delimiter.join(log_msg)
        MB = 1024.0 * 1024.0
        for obj in iterable:
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == len(iterable) - 1:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (len(iterable) - i))))
                kw = dict(eta=eta, meters=str(self), time=str(iter_time), data=str(data_time))
                if torch.cuda.is_available():
                    kw["memory"] = torch.cuda.max_memory_allocated() / MB
                print(log_msg.format(i, len(iterable), **kw))
            i += 1
            end = time.time()
        total_time = time.time() - start_time
        print("{} Total time: {} ({:.4f} s / it)".format(header, str(datetime.timedelta(seconds=int(total_time))),
                                                         total_time / max(1, len(iterable))))


def setup_for_distributed(is_master):
    import builtins as __builtin__
    builtin_print = __builtin__.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    __builtin__.print = print


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def init_distributed_mode(args):
    """utils.py:264-294; backend nccl (= RCCL on ROCm) for cuda, gloo for --device cpu."""
    if getattr(args, "dist_on_itp", False):
        args.rank = int(os.environ["OMPI_COMM_WORLD_RANK"])
        args.world_size = int(os.environ["OMPI_COMM_WORLD_SIZE"])
        args.gpu = int(os.environ["OMPI_COMM_WORLD_LOCAL_RANK"])
        args.dist_url = "tcp://%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"])
        os.environ["LOCAL_RANK"], os.environ["RANK"], os.environ["WORLD_SIZE"] = str(args.gpu), str(args.rank), str(args.world_size)
    elif "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ["LOCAL_RANK"])
    elif "SLURM_PROCID" in os.environ:
        args.rank = int(os.environ["SLURM_PROCID"])
        args.gpu = args.rank % max(1, torch.cuda.device_count())
    else:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.distributed = True
    on_gpu = str(getattr(args, "device", "cuda")).startswith("cuda")
    if on_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if on_gpu else "gloo"
    print("| distributed init (rank {}): {}, gpu {}".format(args.rank, args.dist_url, args.gpu), flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.dist_url, world_size=args.world_size,
                            rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def cleanup_distributed_mode():
    if is_dist_avail_and_initialized():
        dist.destroy_process_group()


class NativeScalerWithGradNormCount:
    """utils.py:351-377.  bf16 needs no loss scaling: backward -> (fused) clip -> step; the returned
    value is the total gradient norm BEFORE clipping, like clip_grad_norm_ / get_grad_norm_.
    ``loss`` is either an autograd tensor or the model's fused-loss handle (``model.backward``)."""
    state_dict_key = "amp_scaler"

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True,
                 model=None, reducer=None, poison=None):
        if model is not None and getattr(model, "_fused_loss_pending", False):
            model.backward()
            model._fused_loss_pending = False
        else:
            loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        if reducer is not None:
            if poison is not None and hasattr(reducer, "sync_flag"):
                reducer.sync_flag(poison)          # the flag is GLOBAL before it gates the update (every rank skips, every rank raises)
            reducer.finish()
        norm = optimizer.engine.grad_norm()
        if poison is not None:
            # a sample of this batch was flagged by the transform chain (the reference raises inside the transform, before
            # the sample reaches the model): a non-finite norm makes the update kernel skip this step on the device
            norm.add_(torch.where(poison > 0, float("nan"), 0.0).to(norm.dtype))
        optimizer.max_norm = float(clip_grad) if clip_grad else 0.0
        optimizer.step()
        return norm

    def state_dict(self):
        return {"scale": 1.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                "_growth_tracker": 0}

    def load_state_dict(self, state_dict):
        pass


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """utils.py:380-392 (generic helper, torch plumbing)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    if len(parameters) == 0:
        return torch.tensor(0.0)
    device = parameters[0].grad.device
    if norm_type == math.inf:
        return max(p.grad.detach().abs().max().to(device) for p in parameters)
    return torch.norm(torch.stack([torch.norm(p.grad.detach(), norm_type).to(device) for p in parameters]), norm_type)


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0,
                     warmup_steps=-1):
    """utils.py:395-412."""
    warmup_schedule = np.array([])
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    print("Set warmup steps = %d" % warmup_iters)
    if warmup_epochs > 0:
        warmup_schedule = np.linspace(start_warmup_value, base_value, warmup_iters)
    iters = np.arange(epochs * niter_per_ep - warmup_iters)
    schedule = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / (len(iters))))
                         for i in iters])
    schedule = np.concatenate((warmup_schedule, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """utils.py:425-447 (torch.amp branch): same file name and keys."""
    output_dir = Path(args.output_dir)
    to_save = {"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
               "scaler": loss_scaler.state_dict(), "args": args}
    # additions to the reference's keys (its loaders index the five above and ignore the rest): the numerics switches the run
    # was trained with (tokenizer mode, stored GELU derivative, precision), and the state of the model's drop-path generator
    to_save["numerics"] = getattr(args, "numerics", None)
    # (every rank has its OWN stream, seeded seed + rank: the checkpoint holds one state per rank -- all ranks call save_model,
    # as in the reference, so the gather below is a collective every rank takes part in)
    dps = getattr(model_without_ddp, "_dp_stream", None)
    if dps is not None:
        to_save["drop_path_rng"] = gather_rank_states(dps.state())
    save_on_master(to_save, output_dir / ("checkpoint-%s.pth" % str(epoch)))


def gather_rank_states(state):
    """{'world': W, 'states': [uint8 tensor of rank 0, ..., of rank W - 1]} from each rank's generator state."""
    state = torch.as_tensor(state, dtype=torch.uint8).cpu()
    world = get_world_size()
    if world == 1:
        return {"world": 1, "states": [state]}
    states = [None] * world
    dist.all_gather_object(states, state)
    return {"world": world, "states": [torch.as_tensor(s, dtype=torch.uint8).cpu() for s in states]}


def restore_rank_state(stream, saved):
    """Rank r continues ITS OWN stream: states[r] when the checkpoint was written by a job of the same world size.  A
    checkpoint of another world size (or one of the round-5 format: a single state, rank 0's) has no state for this rank's
    stream -- the stream keeps the `seed + rank` the entrypoint has just given it (what the reference does on every resume:
    it re-seeds per rank), and rank 0 says so.  Returns True when a saved state was loaded."""
    if isinstance(saved, dict) and "states" in saved:
        if int(saved.get("world", len(saved["states"]))) == get_world_size() and get_rank() < len(saved["states"]):
            stream.load_state(saved["states"][get_rank()])
            return True
        print("drop-path streams: checkpoint of world size %s, this job %d -- every rank restarts its stream from seed + rank"
              % (saved.get("world"), get_world_size()))
        return False
    if get_world_size() == 1:                              # round-5 checkpoints: one state, rank 0's
        stream.load_state(saved)
        return True
    print("drop-path streams: single-state checkpoint in a %d-rank job -- every rank restarts its stream from seed + rank"
          % get_world_size())
    return False


def auto_load_model(args, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """utils.py:485-519."""
    output_dir = Path(args.output_dir)
    if args.auto_resume and len(args.resume) <= 1:
        latest = -1
        for ckpt in glob.glob(os.path.join(output_dir, "checkpoint-*.pth")):
            t = ckpt.split("-")[-1].split(".")[0]
            if t.isdigit():
                latest = max(int(t), latest)
        if latest >= 0:
            args.resume = os.path.join(output_dir, "checkpoint-%d.pth" % latest)
        print("Auto resume checkpoint: %s" % args.resume)
    print(f"Resuming from {args.resume}")
    if args.resume:
        checkpoint = torch.load(args.resume, map_location="cpu", weights_only=False)
        model_without_ddp.load_state_dict(checkpoint["model"], strict=True)
        print("Resume checkpoint %s" % args.resume)
        if optimizer is not None and "optimizer" in checkpoint and "epoch" in checkpoint:
            optimizer.load_state_dict(checkpoint["optimizer"])
            epoch = checkpoint["epoch"] if checkpoint["epoch"] != "best" else args.epochs
            args.start_epoch = epoch + 1
            if "scaler" in checkpoint:
                loss_scaler.load_state_dict(checkpoint["scaler"])
            if "drop_path_rng" in checkpoint and hasattr(model_without_ddp, "_dp_stream"):
                restore_rank_state(model_without_ddp._dp_stream, checkpoint["drop_path_rng"])
            was = checkpoint.get("numerics")
            now = getattr(args, "numerics", None)
            if was is not None and now is not None and was != now:
                print(f"WARNING: checkpoint was trained with numerics {was}, this run uses {now}")
            print("With optim & sched!")


def load_state_dict(model, state_dict, prefix="", ignore_missing="relative_position_index"):
    """utils.py:302-348: non-strict load, module by module through ``_load_from_state_dict`` (so buffers / parameters
    missing on either side are collected instead of raised), with the reference's four report lines; missing keys
    that contain one of the ``|``-separated ``ignore_missing`` patterns are listed separately."""
    missing, unexpected, errors = [], [], []
    metadata = getattr(state_dict, "_metadata", None)
    sd = state_dict.copy()
    if metadata is not None:
        sd._metadata = metadata
    stack = [(model, prefix)]
    while stack:                                                   # pre-order, children in registration order
        module, pre = stack.pop()
        module._load_from_state_dict(sd, pre, {} if metadata is None else metadata.get(pre[:-1], {}), True, missing,
                                     unexpected, errors)
        stack.extend(reversed([(child, pre + name + ".") for name, child in module._modules.items() if child is not None]))
    patterns = ignore_missing.split("|")
    ignored = [k for k in missing if any(pat in k for pat in patterns)]
    warned = [k for k in missing if k not in ignored]
    cls = model.__class__.__name__
    for keys, text in ((warned, "Weights of {} not initialized from pretrained model: {}"),
                       (unexpected, "Weights from pretrained model not used in {}: {}"),
                       (ignored, "Ignored weights of {} not initialized from pretrained model: {}")):
        if keys:
            print(text.format(cls, keys))
    if errors:
        print("\n".join(errors))
    if getattr(model, "_engine", None) is not None:
        model._engine.weights_dirty = True


def _resize_pos_embed(pos_embed_checkpoint, model):
    """Bicubic resize of the patch part of an absolute position embedding to the model's grid (utils.py:703-721)."""
    dim = pos_embed_checkpoint.shape[-1]
    n_extra = model.pos_embed.shape[-2] - model.patch_embed.num_patches
    src = int((pos_embed_checkpoint.shape[-2] - n_extra) ** 0.5)
    dst = int(model.patch_embed.num_patches ** 0.5)
    if src == dst:
        return pos_embed_checkpoint
    print("Position interpolate from %dx%d to %dx%d" % (src, src, dst, dst))
    grid = pos_embed_checkpoint[:, n_extra:].reshape(-1, src, src, dim).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(dst, dst), mode="bicubic", align_corners=False)
    return torch.cat((pos_embed_checkpoint[:, :n_extra], grid.permute(0, 2, 3, 1).flatten(1, 2)), dim=1)


def _resample_rel_pos_table(table, src_size, dst_size, num_extra_tokens):
    """utils.py:657-699: relative-position tables of another window size.  Source positions follow a geometric
    progression (bisection of the ratio so that the progression spans the target half-width), every head's
    src x src grid is resampled by a bicubic spline at the integer target positions, the extra (cls) rows are kept.
    The reference calls scipy.interpolate.interp2d(x, y, z, kind='cubic'), removed in SciPy 1.14; on a regular grid
    that function was FITPACK's regrid_smth(kx = ky = 3, s = 0) -- the call RectBivariateSpline makes (third party,
    parity unpinned: the reference's own call cannot run on this SciPy)."""
    import numpy as np
    from scipy.interpolate import RectBivariateSpline
    extra = table[-num_extra_tokens:, :]
    body = table[:-num_extra_tokens, :]

    def geometric_progression(a, r, n):
        return a * (1.0 - r ** n) / (1.0 - r)
    left, right = 1.01, 1.5
    while right - left > 1e-6:
        q = (left + right) / 2.0
        if geometric_progression(1, q, src_size // 2) > dst_size // 2:
            right = q
        else:
            left = q
    dis, cur = [], 1
    for i in range(src_size // 2):
        dis.append(cur)
        cur += q ** (i + 1)
    x = [-v for v in reversed(dis)] + [0] + dis
    t = dst_size // 2.0
    dx = np.arange(-t, t + 0.1, 1.0)
    print("Original positions = %s" % str(x))
    print("Target positions = %s" % str(dx))
    heads = []
    for i in range(body.shape[1]):
        z = body[:, i].view(src_size, src_size).float().numpy()           # z[iy, ix]
        f = RectBivariateSpline(x, x, z.T.astype(np.float64), kx=3, ky=3, s=0)
        heads.append(torch.Tensor(f(dx, dx).T.copy()).contiguous().view(-1, 1).to(table.device))   # interp2d returns [len(y), len(x)]
    return torch.cat((torch.cat(heads, dim=-1), extra), dim=0)


def finetune(args, model):
    """utils.py:613-723: initialise a finetuning model from a pretraining checkpoint -- pick the state dict by
    ``args.model_key``, drop a head of another shape, EXPAND the shared relative-position table to every block when
    the model keeps per-block tables, drop the index buffers, then the non-strict load.  The two resampling
    branches apply only when the finetuning resolution differs from the pretraining one: pos_embed is resized
    (bicubic), the bias tables are resampled on the geometric-progression grid (_resample_rel_pos_table)."""
    checkpoint = torch.load(args.finetune, map_location="cpu", weights_only=False)
    print("Load ckpt from %s" % args.finetune)
    ckpt = checkpoint
    for model_key in getattr(args, "model_key", "model|module").split("|"):
        if model_key in checkpoint:
            ckpt = checkpoint[model_key]
            print("Load state_dict by model_key = %s" % model_key)
            break
    own = model.state_dict()
    for k in ("head.weight", "head.bias"):
        if k in ckpt and ckpt[k].shape != own[k].shape:
            print(f"Removing key {k} from pretrained checkpoint")
            del ckpt[k]
    shared = "rel_pos_bias.relative_position_bias_table"
    if model.use_rel_pos_bias and shared in ckpt:
        print("Expand the shared relative position embedding to each transformer block. ")
        table = ckpt.pop(shared)
        for i in range(model.get_num_layers()):
            ckpt["blocks.%d.attn.relative_position_bias_table" % i] = table.clone()
    for key in [k for k in ckpt if "relative_position_index" in k]:
        ckpt.pop(key)
    for key in [k for k in ckpt if "relative_position_bias_table" in k and k in own]:
        src_num_pos, dst_num_pos = ckpt[key].shape[0], own[key].shape[0]
        ps = model.patch_embed.patch_shape
        num_extra = dst_num_pos - (ps[0] * 2 - 1) * (ps[1] * 2 - 1)
        src_size, dst_size = int((src_num_pos - num_extra) ** 0.5), int((dst_num_pos - num_extra) ** 0.5)
        print(ps, src_size, dst_size)
        if src_size != dst_size:
            print("Position interpolate for %s from %dx%d to %dx%d" % (key, src_size, src_size, dst_size, dst_size))
            ckpt[key] = _resample_rel_pos_table(ckpt[key], src_size, dst_size, num_extra)
    if "pos_embed" in ckpt and model.pos_embed is not None:
        ckpt["pos_embed"] = _resize_pos_embed(ckpt["pos_embed"], model)
    load_state_dict(model, ckpt, prefix=getattr(args, "model_prefix", ""))


def create_d_vae(weight_path, d_vae_type, image_size, device):
    if d_vae_type == "event":
        return get_event_vae(weight_path, image_size, device)
    raise NotImplementedError()


def get_event_vae(weight_path, image_size, device):
    """utils.py:571-578: checkpoint {'hparams', 'weights', ...} written by eventvae/train_vae.py:271-290."""
    from .vae_model import DiscreteVAE
    loaded_obj = torch.load(weight_path, map_location="cpu", weights_only=False)
    vae_params, weights = loaded_obj["hparams"], loaded_obj["weights"]
    vae = DiscreteVAE(**vae_params).to(device)
    vae.load_state_dict(weights)
    print(f"loaded event vae from {weight_path}")
    return vae


def host_cpu_budget():
    """CPUs this process may actually use: the scheduler affinity capped by the container's CFS quota (cgroup v2
    cpu.max / v1 cpu.cfs_quota_us).  os.cpu_count() reports the machine's cores (256 on the GPU box) although the
    container is throttled at 16: threads beyond the quota only burn it."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cap_host_threads(limit=4):
    """The training loop's host work is launch enqueueing plus tiny index arithmetic; torch's default intra-op pool (one
    thread per visible core, spinning after every parallel region) exhausts the container's CPU quota and the launch
    thread gets throttled for the rest of the CFS period (measured on the GPU box: 76 s of CPU in a 10 s run, 80 ms
    stalls of the launch thread, GPU idle).  Caps the pool at min(limit, quota); never raises it."""
    ranks_here = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))        # torchrun: rank processes sharing this quota
    n = max(1, min(limit, host_cpu_budget() // ranks_here, torch.get_num_threads()))
    if n < torch.get_num_threads():
        torch.set_num_threads(n)
    return n


class DropPathStream:
    """The model's own stochastic-depth generator (modeling_pretrain / modeling_finetune `_dp_uniform`): a CPU torch.Generator
    that no other consumer of the global stream shifts.  Seeded EXPLICITLY by the entrypoint (`seed(args.seed + rank)`), lazily
    from torch's seed of the moment otherwise; `state()` / `load_state()` travel with the checkpoint, ONE STATE PER RANK
    (utils.save_model / auto_load_model: key "drop_path_rng" = {world, states[rank]}), so every rank of a resumed run continues its
    own mask stream (and the replicas' masks stay decorrelated) instead of replaying it from step 0.
    `uniform(rows, B, device)` uploads through a pinned ring (HostStager): no host-blocking copy on the launch stream."""

    def __init__(self):
        self.gen = None
        self._stager = None

    def _g(self):
        if self.gen is None:
            self.seed(int(torch.initial_seed()))
        return self.gen

    def seed(self, seed):
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed((int(seed) ^ 0x5DEECE66D) & ((1 << 63) - 1))

    def state(self):
        return self._g().get_state().clone()

    def load_state(self, st):
        self._g().set_state(torch.as_tensor(st, dtype=torch.uint8).cpu())

    def uniform(self, rows, B, device=None):
        u = torch.rand((rows, B), generator=self._g())
        if device is None:
            return u
        n = rows * B * 4
        if self._stager is None or self._stager.host[0].numel() < n:
            self._stager = HostStager(n, device)
        return self._stager.put(u.numpy()).view(rows, B)


class HostStager:
    """Small host -> device uploads without stalling the launch queue.

    ``tensor.cuda()`` from pageable memory blocks the host until the copy has run, i.e. until every
    kernel queued before it has finished: one per training step is enough to drain the queue and
    expose launch latency on the kernels that follow.  The stager keeps a ring of pinned host
    buffers + device twins; ``put`` copies into the next pinned slot and issues an asynchronous
    stream-ordered upload (an event per slot guards reuse)."""

    def __init__(self, nbytes, device, slots=4):
        import torch
        self.host = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.dev = [torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.ev = [None] * slots
        self.k = 0

    def put(self, array):
        """array: contiguous numpy array; returns a device tensor view (same dtype, flat) valid until
        `slots` further calls."""
        import numpy as np
        import torch
        k = self.k
        self.k = (k + 1) % len(self.host)
        if self.ev[k] is not None:
            self.ev[k].synchronize()
        flat = np.ascontiguousarray(array).reshape(-1)
        n = flat.nbytes
        assert n <= self.host[k].numel(), "HostStager slot too small"
        self.host[k][:n].numpy()[:] = flat.view(np.uint8)
        self.dev[k][:n].copy_(self.host[k][:n], non_blocking=True)
        # blocking=True: a host that is a full ring ahead of the GPU SLEEPS in synchronize() instead of spinning (the launch
        # thread shares the container's CPU quota with the DataLoader workers and, on a multi-GPU node, the other ranks)
        e = torch.cuda.Event(blocking=os.environ.get("MEMHIP_STAGER_SPIN") != "1")
        e.record()
        self.ev[k] = e
        return self.dev[k][:n].view(torch.from_numpy(flat[:0]).dtype)
