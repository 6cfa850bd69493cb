"""One data-parallel rank of the tiny pretraining job used by tests/test_ddp_gpu.py (run as a subprocess):
the product's own N>1 path -- init_process_group, GradReducer as the engine's grad_hook (parameter broadcast from
rank 0, one async all-reduce per bucket from inside backward, join before grad-norm / AdamW) -- on this rank's slice
of a fixed global batch.  Backend gloo on CUDA tensors: both ranks share the ONE GPU of the test box (RCCL refuses two
ranks on one device); the collective's arithmetic (SUM then / world) is the same.
Mirrors /root/reference/mem/run_mem_pretraining.py:302-309 (DistributedSampler split), :365-367 (DDP)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_job(world, rank, steps, per_rank, seed_w):
    """Model + optimizer + this rank's batches.  rank=None: the single-process job on the whole global batch."""
    import contextlib
    import io
    import torch
    from oracle.gen_golden import TINY, vit_inputs
    from oracle.vit_ref import fill_by_name
    from mem_amd.modeling_pretrain import pt_vit
    from mem_amd.optim_factory import create_optimizer

    class A:
        opt = "adamw"; weight_decay = 0.05; lr = 1e-3; opt_eps = 1e-8; opt_betas = [0.9, 0.999]; momentum = 0.9
    cfg = dict(TINY, drop_path_rate=0.1)
    m = pt_vit(**cfg)
    # every rank starts from DIFFERENT weights except rank 0 / the single-process job: the broadcast must fix that
    m.load_state_dict(fill_by_name(m.state_dict(), seed=seed_w + (rank or 0)))
    m = m.cuda().train()
    with contextlib.redirect_stdout(io.StringIO()):
        opt = create_optimizer(A(), m)
    depth = len(m.blocks)
    batches = []
    for it in range(steps):
        parts = [vit_inputs(cfg, per_rank, 4000 + 10 * it + r, 6) for r in range(world)]      # equal M on every rank
        g = torch.Generator().manual_seed(77 + it)
        dp = torch.floor(0.9 + torch.rand((2 * depth, world * per_rank), generator=g))         # fixed keep masks
        if rank is None:
            x = torch.cat([p[0] for p in parts]); mask = torch.cat([p[1] for p in parts]); lab = torch.cat([p[2] for p in parts])
            batches.append((x, mask, lab, dp))
        else:
            x, mask, lab = parts[rank]
            batches.append((x, mask, lab, dp[:, rank * per_rank:(rank + 1) * per_rank].contiguous()))
    return m, opt, batches


def run_steps(m, opt, batches, reducer=None, clip=1.0):
    from mem_amd.utils import NativeScalerWithGradNormCount
    scaler = NativeScalerWithGradNormCount()
    rec = []
    for x, mask, lab, dp in batches:
        la = m.forward_loss(x.cuda(), mask.cuda(), lab.cuda(), drop_path_masks=dp.cuda())
        m._fused_loss_pending = True
        gn = scaler(la, opt, clip_grad=clip, parameters=m.parameters(), model=m, reducer=reducer)
        rec.append((float(la[0]), float(gn)))
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int); ap.add_argument("--world", type=int); ap.add_argument("--port", type=int)
    ap.add_argument("--steps", type=int, default=3); ap.add_argument("--per-rank", type=int, default=3)
    ap.add_argument("--out")
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    from mem_amd.parallel import GradReducer
    m, opt, batches = make_job(a.world, a.rank, a.steps, a.per_rank, seed_w=3)
    eng = m.engine
    red = GradReducer(eng.flat_g, eng.buckets, flat_p=eng.flat_p)
    eng.grad_hook = red
    eng.weights_dirty = True
    rec = run_steps(m, opt, batches, reducer=red)
    torch.cuda.synchronize()
    torch.save({"rec": rec, "flat_p": eng.flat_p.cpu()}, a.out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
