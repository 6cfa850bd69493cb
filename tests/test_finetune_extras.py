"""f3 leftovers: relative-position-table resampling for another window size
(mem/utils.py:657-699) and gradient accumulation in the finetuning loop (mem/engine_for_finetuning.py:78,117-124)."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch


def test_rel_pos_table_resampling_properties():
    """-m "not gpu".  The reference's scipy.interpolate.interp2d(kind='cubic') call cannot run on this SciPy (removed in
    1.14); its regular-grid arithmetic is FITPACK's interpolating bicubic spline, so: (1) the source positions are the
    reference's geometric progression (symmetric, innermost step 1, outermost position >= the target half-width),
    (2) a surface that is a cubic polynomial in x and in y is reproduced EXACTLY at the target positions (interpolating
    cubic splines are exact on cubics), (3) the extra (cls) rows are carried over unchanged, (4) heads are independent."""
    from mem_amd import utils as U
    src, dst, heads = 27, 39, 3
    buf = io.StringIO()
    # recover the source positions the function prints (they are part of the reference's behaviour)
    with contextlib.redirect_stdout(buf):
        U._resample_rel_pos_table(torch.zeros(src * src + 3, heads), src, dst, 3)
    line = [l for l in buf.getvalue().splitlines() if l.startswith("Original positions")][0]
    x = np.array(eval(line.split("=", 1)[1]))
    assert len(x) == src and np.allclose(x, -x[::-1]) and x[src // 2] == 0 and abs(x[src // 2 + 1] - 1) < 1e-12
    assert x[-1] >= dst // 2 - 1e-3 and np.all(np.diff(x) > 0)
    X, Y = np.meshgrid(x, x)                                    # z[iy, ix]
    polys = [0.3 + 0.1 * X - 0.02 * Y + 0.004 * X * Y + 1e-4 * X ** 3 - 2e-4 * Y ** 3 + 3e-5 * X ** 2 * Y,
             1.0 - 0.05 * X ** 2 + 0.01 * Y ** 2, 0.5 * np.ones_like(X)]
    table = torch.zeros(src * src + 3, heads)
    for h, z in enumerate(polys):
        table[: src * src, h] = torch.from_numpy(z.reshape(-1)).float()
    table[-3:] = torch.tensor([[1., 2., 3.], [4., 5., 6.], [7., 8., 9.]])
    with contextlib.redirect_stdout(io.StringIO()):
        out = U._resample_rel_pos_table(table, src, dst, 3)
    assert out.shape == (dst * dst + 3, heads) and torch.equal(out[-3:], table[-3:])
    t = dst // 2
    d = np.arange(-t, t + 0.1, 1.0)
    DX, DY = np.meshgrid(d, d)
    want = [0.3 + 0.1 * DX - 0.02 * DY + 0.004 * DX * DY + 1e-4 * DX ** 3 - 2e-4 * DY ** 3 + 3e-5 * DX ** 2 * DY,
            1.0 - 0.05 * DX ** 2 + 0.01 * DY ** 2, 0.5 * np.ones_like(DX)]
    for h in range(heads):
        got = out[: dst * dst, h].numpy().reshape(dst, dst)
        assert np.abs(got - want[h]).max() <= 2e-5 * max(1.0, np.abs(want[h]).max()), h


@pytest.mark.gpu
def test_finetune_checkpoint_at_another_resolution(tmp_path, capsys):
    """A pretraining checkpoint at 64x96 (window 4x6) initialises a finetuning model at 96x144 (window 6x9): shared table
    expanded to every block AND resampled (27+3... entries), forward / backward run."""
    from mem_amd import utils as U
    from mem_amd.modeling_finetune import ft_vit
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.vit_ref import fill_by_name
    geo = dict(patch_size=(16, 16), in_chans=3, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4)
    pre = pt_vit(img_size=(64, 64), vocab_size=512, drop_path_rate=0.0, use_shared_rel_pos_bias=True, use_abs_pos_emb=False,
                 init_values=0.1, **geo)
    ckpt = os.path.join(tmp_path, "checkpoint-0.pth")
    torch.save({"model": fill_by_name(pre.state_dict(), seed=2)}, ckpt)
    m = ft_vit(img_size=(96, 96), num_classes=3, drop_path_rate=0.0, init_values=0.1, use_abs_pos_emb=False,
               use_rel_pos_bias=True, use_mean_pooling=True, **geo)

    class Args:
        finetune = ckpt; model_key = "model|module"; model_prefix = ""
    U.finetune(Args(), m)
    out = capsys.readouterr().out
    assert "Position interpolate for blocks.0.attn.relative_position_bias_table from 7x7 to 11x11" in out
    t = m.state_dict()["blocks.1.attn.relative_position_bias_table"]
    assert t.shape == (11 * 11 + 3, 2) and torch.isfinite(t).all()
    src = torch.load(ckpt, weights_only=False)["model"]["rel_pos_bias.relative_position_bias_table"]
    assert torch.equal(t[-3:], src[-3:])                       # cls rows carried over
    m = m.cuda().train()
    y = m(torch.rand(2, 3, 96, 96).cuda())
    y.float().sum().backward()
    assert torch.isfinite(y).all()


@pytest.mark.gpu
def test_update_freq_accumulates_like_one_large_batch():
    """update_freq = 2 over two micro-batches == one step on their concatenation (mean CE, equal micro-batch sizes):
    gradients (before the optimizer step) and the parameters after the step, within bf16 tolerance."""
    from mem_amd import engine_for_finetuning as EF
    from mem_amd import optim_factory as OF
    from mem_amd.modeling_finetune import ft_vit
    from mem_amd.utils import NativeScalerWithGradNormCount
    from oracle.vit_ref import fill_by_name
    geo = dict(img_size=(64, 64), patch_size=(16, 16), in_chans=3, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4)

    def make():
        torch.manual_seed(0)
        m = ft_vit(num_classes=4, drop_path_rate=0.0, init_values=0.1, use_abs_pos_emb=True, use_rel_pos_bias=True,
                   use_mean_pooling=True, **geo)
        m.load_state_dict(fill_by_name(m.state_dict(), seed=4))
        m = m.cuda()

        class A:
            opt = "adamw"; weight_decay = 0.05; lr = 1e-3; opt_eps = 1e-8; opt_betas = [0.9, 0.999]; momentum = 0.9
        with contextlib.redirect_stdout(io.StringIO()):
            opt = OF.create_optimizer(A(), m)
        return m, opt
    g = torch.Generator().manual_seed(1)
    x = torch.rand(8, 3, 64, 64, generator=g)
    y = torch.randint(0, 4, (8,), generator=g)
    crit = torch.nn.CrossEntropyLoss()
    m1, o1 = make()
    m2, o2 = make()
    p0 = m1.engine.flat_p.clone()
    grads = {}

    def snapshot_before_step(tag, m, o):                      # the gradient the optimizer actually consumes
        step = o.step

        def wrapped(*a, **k):
            grads[tag] = m.engine.flat_g.clone()
            return step(*a, **k)
        o.step = wrapped
    snapshot_before_step("one", m1, o1)
    snapshot_before_step("acc", m2, o2)
    with contextlib.redirect_stdout(io.StringIO()):
        EF.train_one_epoch(None, m1, crit, [(x, y)], o1, torch.device("cuda"), 0, NativeScalerWithGradNormCount(), 0,
                           lr_schedule_values=[1e-3], update_freq=1)
        EF.train_one_epoch(None, m2, crit, [(x[:4], y[:4]), (x[4:], y[4:])], o2, torch.device("cuda"), 0,
                           NativeScalerWithGradNormCount(), 0, lr_schedule_values=[1e-3], update_freq=2)
    assert not m2.engine.accumulate_grads
    d1, d2 = m1.engine.flat_p - p0, m2.engine.flat_p - p0
    rel = float((d1 - d2).norm() / d1.norm())
    print("relative difference of the parameter update, update_freq 2 vs one batch: %.3e" % rel)
    assert o2.steps == 1 and rel <= 3e-2
    # the first AdamW step is ~ lr * sign(g): it cannot see a mis-scaled accumulated gradient (a missing 1 / update_freq,
    # a double-counted layer-scale gradient ...).  So compare the gradient buffers themselves, magnitude included.
    g1, g2 = grads["one"], grads["acc"]
    rel_g = float((g1 - g2).norm() / g1.norm())
    cos = float(torch.dot(g1, g2) / (g1.norm() * g2.norm()))
    ratio = float(g2.norm() / g1.norm())
    print("accumulated vs one-batch gradient: rel-L2 %.3e, cosine %.6f, norm ratio %.4f" % (rel_g, cos, ratio))
    assert rel_g <= 3e-2 and cos >= 0.9995 and abs(ratio - 1) <= 1e-2
    # per tensor (a wrong factor on ONE small tensor -- e.g. the layer-scale gammas -- hides in the global norm)
    for name, (o, k) in m1.engine.segs.items():
        a, b = g1[o:o + k], g2[o:o + k]
        if float(a.norm()) > 1e-6:
            assert float((a - b).norm() / a.norm()) <= 6e-2, name
