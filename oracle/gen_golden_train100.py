"""tests/golden/vit_base_train100.npz: 100 optimizer steps of ViT-B/16 (C = 2, B = 2) run by the REFERENCE in the build container.

TEST INFRASTRUCTURE.  Needs /root/reference (absent on the GPU box -> exits).  North star: "step-100 pretrain loss within
1e-4 of reference" -- pinned so far at 100 steps on the tiny config and at 10 steps on ViT-B (oracle/gen_golden.py); this is
the ViT-B curve.  Reference model (mem/modeling_pretrain.py pt_vit) + reference optimizer (mem/optim_factory.py
create_optimizer, betas (0.9, 0.95)) + the loop of mem/engine_for_pretraining.py:123-162 restated (oracle.vit_ref.train_step:
the shipped loop cannot run on CPU), clip 30, lr = cosine(5e-4 -> 1e-5, 10 warm-up steps), wd 0.05, four recurring batches.
  set "dp0":  drop_path 0, fp32 (the parity anchor of --precision fp32) and bf16 autocast (what the bf16 engine is measured
              against); the oracle restatement runs beside the reference for the first 10 fp32 steps and must equal it bit for bit;
  set "dp1":  drop_path_rate 0.1, fp32: the keep masks the reference's DropPath drew (timm drop_path, stubbed with its
              published arithmetic in oracle/_refimport.py) are RECORDED and stored, so the product can be fed the same masks.

    python -m oracle.gen_golden_train100        # ~6 min on 8 threads
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refimport as R            # noqa: E402
from oracle import vit_ref as V               # noqa: E402
from oracle.gen_golden import BASE, vit_inputs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
STEPS, NB, NMASK = 100, 2, 98


def batch(cfg, it):
    return vit_inputs(cfg, NB, 2000 + it % 4, NMASK)


def schedules():
    with contextlib.redirect_stdout(io.StringIO()):
        return (V.cosine_scheduler(5e-4, 1e-5, 1, STEPS, warmup_epochs=5, warmup_steps=10),
                V.cosine_scheduler(0.05, 0.05, 1, STEPS))


class OptArgs:
    opt = "adamw"; weight_decay = 0.05; lr = 5e-4; opt_eps = 1e-8; opt_betas = [0.9, 0.999]; momentum = 0.9


def main():
    if not R.install():
        print("no /root/reference here: nothing generated")
        return
    import modeling_finetune as MF
    import modeling_pretrain as MPre
    import optim_factory as OF
    torch.set_num_threads(8)
    lr_s, wd_s = schedules()
    out = {"lr": lr_s, "wd": wd_s}
    t0 = time.time()
    # ---- set dp0: drop_path 0
    cfg = dict(BASE, in_chans=2)
    for mode, dt in (("fp32", None), ("bf16", torch.bfloat16)):
        torch.manual_seed(0)
        ref = MPre.pt_vit(**cfg)
        w = V.fill_by_name(ref.state_dict(), seed=1)
        ref.load_state_dict(w)
        with contextlib.redirect_stdout(io.StringIO()):
            ropt = OF.create_optimizer(OptArgs(), ref)
        ora = oopt = None
        if dt is None:
            ora = V.RefViT(**cfg); ora.load_state_dict(w)
            oopt = V.make_optimizer(ora, lr=5e-4, weight_decay=0.05)
        rec = []
        for it in range(STEPS):
            x, m, l = batch(cfg, it)
            rec.append(V.train_step(ref, ropt, x, m, l, it, lr_s, wd_s, clip_grad=30.0, autocast_dtype=dt))
            if ora is not None and it < 10:
                assert V.train_step(ora, oopt, x, m, l, it, lr_s, wd_s, clip_grad=30.0) == rec[-1], (mode, it)
            if it % 20 == 0:
                print(f"dp0 {mode} step {it}: loss {rec[-1][0]:.6f} gnorm {rec[-1][1]:.4f}  ({time.time() - t0:.0f} s)", flush=True)
        out[f"dp0__{mode}__loss"] = np.array([r[0] for r in rec])
        out[f"dp0__{mode}__gnorm"] = np.array([r[1] for r in rec])
        out[f"dp0__{mode}__acc"] = np.array([r[2] for r in rec])
        if dt is None:
            out["dp0__fp32__final__lm_head.bias"] = ref.lm_head.bias.detach().numpy().copy()
            out["dp0__fp32__final__cls_token"] = ref.cls_token.detach().numpy().copy()
    # ---- set dp1: drop_path_rate 0.1, the reference's own keep masks recorded
    cfg1 = dict(BASE, in_chans=2, drop_path_rate=0.1)
    torch.manual_seed(0)
    ref = MPre.pt_vit(**cfg1)
    ref.load_state_dict(V.fill_by_name(ref.state_dict(), seed=1))
    with contextlib.redirect_stdout(io.StringIO()):
        ropt = OF.create_optimizer(OptArgs(), ref)
    drawn = []
    orig = MF.drop_path

    def recording_drop_path(x, drop_prob=0.0, training=False):
        # (the stub's arithmetic, oracle/_refimport.py: r = floor(keep + U[0,1)); y = x / keep * r) with r kept
        if drop_prob == 0.0 or not training:
            drawn.append(torch.ones(x.shape[0]))
            return x
        keep = 1 - drop_prob
        r = (keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)).floor_()
        drawn.append(r.view(-1).clone())
        return x.div(keep) * r
    MF.drop_path = recording_drop_path
    try:
        rec, masks = [], []
        for it in range(STEPS):
            torch.manual_seed(7000 + it)                     # the draws of step `it`
            del drawn[:]
            x, m, l = batch(cfg1, it)
            rec.append(V.train_step(ref, ropt, x, m, l, it, lr_s, wd_s, clip_grad=30.0))
            # (a block whose rate is 0 -- block 0 of the linspace -- holds nn.Identity, not DropPath: no draw)
            dpr = [v.item() for v in torch.linspace(0, cfg1["drop_path_rate"], cfg1["depth"])]
            assert len(drawn) == 2 * sum(r > 0 for r in dpr), len(drawn)
            rows, k = [], 0
            for r in dpr:
                for _ in range(2):
                    if r > 0:
                        rows.append(drawn[k]); k += 1
                    else:
                        rows.append(torch.ones(NB))
            masks.append(torch.stack(rows).numpy().astype(np.uint8))            # [2 * depth, B]: block i -> rows 2 i (attention), 2 i + 1 (MLP)
            if it % 20 == 0:
                print(f"dp1 fp32 step {it}: loss {rec[-1][0]:.6f}  dropped {int((masks[-1] == 0).sum())} of {masks[-1].size}  ({time.time() - t0:.0f} s)", flush=True)
    finally:
        MF.drop_path = orig
    out["dp1__fp32__loss"] = np.array([r[0] for r in rec])
    out["dp1__fp32__gnorm"] = np.array([r[1] for r in rec])
    out["dp1__keep"] = np.stack(masks)
    np.savez_compressed(os.path.join(OUT, "vit_base_train100.npz"), **out)
    print("wrote vit_base_train100.npz in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
