"""-m gpu: the `--precision fp32` parity mode (mem_amd/vit_engine_f32.py -> csrc/fp32_path.hip: fp32 MFMA GEMMs, fp32
attention / LayerNorm / loss, no bf16 rounding points) against the REFERENCE's fp32 CPU runs (tests/golden, produced by
the imported reference):
  * tiny model, one forward/backward: logits 2e-5, loss 2e-6, every gradient rel-L2 <= 2e-5;
  * tiny model, 100 optimizer steps (reference create_optimizer, clip 30): loss within 1e-4 of the reference at every
    step incl. step 100 (BASELINE north star);
  * BASELINE config #1 (ViT-B/16 C=3, B=2, 10 steps, ncaltech.conf hyper-parameters): per-step loss <= 1e-5 relative
    (SURVEY.md section 8d);
  * ViT-B C=2: loss and the stored gradient tensors.
Reference arithmetic: mem/modeling_pretrain.py:97-126, mem/modeling_finetune.py:56-189, mem/engine_for_pretraining.py:147-162."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _model(cfg, seed):
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.vit_ref import fill_by_name
    m = pt_vit(precision="fp32", **cfg)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=seed))
    return m.cuda().train()


def test_tiny_forward_backward_fp32_vs_reference():
    from test_model_gpu import TINY
    g = np.load(os.path.join(GOLDEN, "vit_tiny_fwdbwd.npz"))
    m = _model(TINY, 0)
    assert type(m.engine).__name__ == "ViTEngineF32"
    x, mask, labels = (torch.from_numpy(g[k]).cuda() for k in ("x", "mask", "labels"))
    la = m.forward_loss(x, mask, labels)
    assert abs(la[0].item() - float(g["fp32__loss"])) <= 2e-6, (la[0].item(), float(g["fp32__loss"]))
    m.eval()
    with torch.no_grad():
        lo = m(x, mask)
    m.train()
    assert lo.dtype == torch.float32 and np.abs(lo.cpu().numpy() - g["fp32__logits"]).max() <= 2e-5
    m.forward_loss(x, mask, labels)
    m.backward()
    worst = 0.0
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g[f"fp32__grad__{k}"]).cuda()
        rel = ((p.grad - ref).norm() / (ref.norm() + 1e-20)).item()
        worst = max(worst, rel)
        assert rel <= 2e-5, (k, rel)
    print("worst gradient rel-L2 vs the reference fp32 run: %.2e" % worst)
    # autograd surface (loss outside the pipeline) gives the same gradients
    m.zero_grad()
    logits = m(x, mask)
    torch.nn.CrossEntropyLoss()(logits, labels).backward()
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g[f"fp32__grad__{k}"]).cuda()
        assert ((p.grad - ref).norm() / (ref.norm() + 1e-20)).item() <= 2e-5, k


def _run(cfg, n_steps, batch_fn, lr, wd, clip, seed_w, precision="fp32", keep=None):
    from mem_amd.modeling_pretrain import pt_vit
    from mem_amd.optim_factory import create_optimizer
    from mem_amd.utils import NativeScalerWithGradNormCount
    from oracle.vit_ref import fill_by_name

    class A:
        opt = "adamw"; weight_decay = 0.05; lr = 5e-4; opt_eps = 1e-8; opt_betas = [0.9, 0.999]; momentum = 0.9
    if precision == "fp32":
        m = _model(cfg, seed_w)
    else:
        m = pt_vit(**cfg)
        m.load_state_dict(fill_by_name(m.state_dict(), seed=seed_w))
        m = m.cuda().train()
    with contextlib.redirect_stdout(io.StringIO()):
        opt = create_optimizer(A(), m)
    scaler = NativeScalerWithGradNormCount()
    rec = []
    for it in range(n_steps):
        for grp in opt.param_groups:
            grp["lr"] = lr[it] * grp["lr_scale"]
            if grp["weight_decay"] > 0:
                grp["weight_decay"] = wd[it]
        x, mask, labels = batch_fn(it)
        dpm = None if keep is None else torch.from_numpy(keep[it].astype(np.float32)).cuda()
        la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda(), drop_path_masks=dpm)
        m._fused_loss_pending = True
        gn = scaler(la, opt, clip_grad=clip, parameters=m.parameters(), model=m)
        rec.append((la[0].item(), gn.item()))
    return np.array(rec)


def test_tiny_100_steps_fp32_within_1e4_of_reference():
    from oracle.gen_golden import TINY, vit_inputs
    g = np.load(os.path.join(GOLDEN, "vit_tiny_train100.npz"))
    rec = _run(TINY, 100, lambda it: vit_inputs(TINY, 4, 1000 + it % 8, 6), g["lr"], g["wd"], 30.0, 0)
    d = np.abs(rec[:, 0] - g["fp32__loss"])
    print("fp32 mode vs reference fp32 curve: max |dloss| %.2e, at step 100 %.2e; grad-norm rel %.2e" %
          (d.max(), d[-1], np.abs(rec[:, 1] / g["fp32__gnorm"] - 1).max()))
    assert d[-1] <= 1e-4 and d.max() <= 1e-4
    assert np.abs(rec[:, 1] / g["fp32__gnorm"] - 1).max() <= 1e-3


def test_config1_vit_base_10_steps_fp32():
    from oracle.gen_golden import BASE, vit_inputs
    from oracle.vit_ref import cosine_scheduler
    g = np.load(os.path.join(GOLDEN, "vit_base_c3.npz"))
    cfg = dict(BASE, in_chans=3)
    wd10 = cosine_scheduler(0.05, 0.05, 1, 10)
    rec = _run(cfg, 10, lambda it: vit_inputs(cfg, 2, 500 + it, 98), g["cfg1__lr"], wd10, 30.0, 1)
    rel = np.abs(rec[:, 0] / g["cfg1__loss"] - 1)
    print("config #1 per-step relative loss difference vs the reference fp32 curve:", rel)
    assert rel.max() <= 1e-5
    assert np.abs(rec[:, 1] / g["cfg1__gnorm"] - 1).max() <= 1e-3


def test_vit_base_c2_gradients_fp32():
    from oracle.gen_golden import BASE, BASE_GRAD_SAMPLES, vit_inputs
    g = np.load(os.path.join(GOLDEN, "vit_base_c2.npz"))
    cfg = dict(BASE, in_chans=2)
    m = _model(cfg, 1)
    x, mask, labels = vit_inputs(cfg, 2, 77, 98)
    la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
    assert abs(la[0].item() / float(g["fp32__loss"]) - 1) <= 2e-6
    m.backward()
    pd = dict(m.named_parameters())
    for k, st in BASE_GRAD_SAMPLES:
        r = torch.from_numpy(g[f"fp32__grad__{k}"]).cuda()
        q = pd[k].grad[::st]
        rel = ((q - r).norm() / (r.norm() + 1e-20)).item()
        assert rel <= 5e-5, (k, rel)


# ---- ViT-B at 100 steps (round 6; oracle/gen_golden_train100.py -> tests/golden/vit_base_train100.npz: the REFERENCE model + the
# reference optimizer, C = 2, B = 2, clip 30, four recurring batches).  North star: step-100 loss within 1e-4 of the reference.
def _train100(precision, which, drop_path_rate=0.0):
    from oracle.gen_golden import BASE, vit_inputs
    g = np.load(os.path.join(GOLDEN, "vit_base_train100.npz"))
    cfg = dict(BASE, in_chans=2, drop_path_rate=drop_path_rate)
    keep = g["dp1__keep"] if which == "dp1" else None
    rec = _run(cfg, 100, lambda it: vit_inputs(cfg, 2, 2000 + it % 4, 98), g["lr"], g["wd"], 30.0, 1, precision=precision, keep=keep)
    return rec, g


def test_vit_base_100_steps_fp32_within_1e4_of_reference():
    rec, g = _train100("fp32", "dp0")
    d = np.abs(rec[:, 0] - g["dp0__fp32__loss"])
    print("ViT-B fp32 mode vs the reference's fp32 curve: max |dloss| %.2e, at step 100 %.2e; grad-norm rel %.2e" %
          (d.max(), d[-1], np.abs(rec[:, 1] / g["dp0__fp32__gnorm"] - 1).max()), "worst step", int(d.argmax()) + 1)
    # the north star's bar is the loss at step 100; along the way two fp32 trajectories (other summation orders) drift apart and
    # re-converge while the loss falls from 9.5 to 4.8 -- measured worst 2.6e-4 (mid-curve), stated here at 1e-3
    assert d[-1] <= 1e-4 and d[:10].max() <= 1e-5 and d.max() <= 1e-3
    assert np.abs(rec[:, 1] / g["dp0__fp32__gnorm"] - 1).max() <= 5e-3


def test_vit_base_100_steps_fp32_with_the_references_drop_path_masks():
    """drop_path_rate 0.1: the keep masks the reference drew (recorded by the generator) are fed to the product."""
    rec, g = _train100("fp32", "dp1", drop_path_rate=0.1)
    assert int((g["dp1__keep"] == 0).sum()) > 50                    # the masks do drop samples
    d = np.abs(rec[:, 0] - g["dp1__fp32__loss"])
    print("ViT-B fp32 mode, reference drop-path masks: max |dloss| %.2e, at step 100 %.2e" % (d.max(), d[-1]))
    assert d[-1] <= 1e-4 and d[:10].max() <= 1e-5 and d.max() <= 1e-3
    assert np.abs(rec[:, 1] / g["dp1__fp32__gnorm"] - 1).max() <= 5e-3


def test_vit_base_100_steps_bf16_engine_deviation():
    """The bf16 product engine on the same 100 steps: its distance from the reference's bf16-autocast curve and from the fp32
    curve, next to the reference's own autocast-vs-fp32 distance (bf16 operands: accumulation-order noise, not 1e-4)."""
    rec, g = _train100("bf16", "dp0")
    d16 = np.abs(rec[:, 0] - g["dp0__bf16__loss"])
    d32 = np.abs(rec[:, 0] - g["dp0__fp32__loss"])
    own = np.abs(g["dp0__bf16__loss"] - g["dp0__fp32__loss"])
    print("ViT-B bf16 engine vs reference bf16 curve: max %.2e, step 100 %.2e | vs fp32 curve: max %.2e, step 100 %.2e | "
          "reference autocast vs its fp32 run: max %.2e, step 100 %.2e" % (d16.max(), d16[-1], d32.max(), d32[-1], own.max(), own[-1]))
    # measured (round 6, two boxes): the product's bf16 engine ends 1.8e-3 / 1.4e-2 from the reference's FP32 curve (worst step
    # 4.3e-2 / 2.4e-1: the curve falls from 9.5 to 4.8 on four recurring batches, bf16 rounding noise is amplified and the
    # engine's atomics make two runs differ), while the reference's own bf16-autocast CPU run ends 3.9e-2 from its fp32 run
    # (worst 3.3e-1).  So the bar is relative: the engine must be at least as close to the fp32 curve as the reference's
    # autocast run is, at step 100 and at the worst step; the first ten steps (before the noise is amplified) are tight.
    assert d32[:10].max() <= 5e-3, d32[:10]
    assert d32[-1] <= own[-1] and d32.max() <= own.max()
    assert d16[-1] <= 2.0 * own[-1]             # (and it ends within twice that distance of the autocast curve itself)
