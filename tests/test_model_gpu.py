"""-m gpu: the fused HIP ViT (mem_amd.modeling_pretrain) against the reference goldens
(tests/golden/vit_*.npz, produced by the imported reference on CPU) and the oracle.

Tolerances (stated, per dtype): the product computes like the reference under bf16 autocast
(bf16 GEMM operands, fp32 accumulate); against the reference's own bf16-autocast CPU run the
remaining differences are accumulation order and a few fused roundings:
  logits  |d| <= 0.03 abs (bf16 resolution at |logit|~2-4 is 0.016)
  loss    |d| <= 2e-3      (vs bf16 golden)   /  <= 1e-2 (vs fp32 golden)
  grads   relative L2 error per tensor <= 3e-2 (vs bf16 golden), global cosine >= 0.999
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TINY = dict(img_size=(64, 64), patch_size=(16, 16), in_chans=3, vocab_size=512, embed_dim=128, depth=2,
            num_heads=2, mlp_ratio=4, drop_path_rate=0.0, use_shared_rel_pos_bias=True,
            use_abs_pos_emb=False, init_values=0.1)


def _tiny_model():
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.vit_ref import fill_by_name
    m = pt_vit(**TINY)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=0))
    return m.cuda().train()


def test_state_dict_surface_and_init_parity():
    """Same keys/shapes as the reference and -- same torch seed -- the same initial weights."""
    from mem_amd.modeling_pretrain import create_model
    from oracle.vit_ref import RefViT
    meta = json.load(open(os.path.join(GOLDEN, "vit_meta.json")))
    torch.manual_seed(0)
    m = create_model("pt_vit", pretrained=False, drop_block_rate=None, **TINY)
    assert list(m.state_dict().keys()) == meta["tiny_state_keys"]
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == meta["tiny_state_shapes"]
    torch.manual_seed(0)
    o = RefViT(**TINY)
    for (k, a), (_, b) in zip(m.state_dict().items(), o.state_dict().items()):
        assert torch.equal(a, b), k
    assert m.get_num_layers() == 2 and m.no_weight_decay() == {"pos_embed", "cls_token"}
    assert m.patch_embed.patch_size == (16, 16) and m.patch_embed.patch_shape == (4, 4)


def test_tiny_forward_backward_vs_reference_golden():
    g = np.load(os.path.join(GOLDEN, "vit_tiny_fwdbwd.npz"))
    m = _tiny_model()
    x, mask, labels = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["mask"]).cuda(), torch.from_numpy(g["labels"]).cuda()
    # --- API path: logits with autograd, loss outside (like the reference engine does)
    logits = m(x, mask)
    assert logits.shape == tuple(g["bf16__logits"].shape)
    d = (logits.detach().float().cpu().numpy() - g["bf16__logits"])
    assert np.abs(d).max() <= 0.03, np.abs(d).max()
    loss = torch.nn.CrossEntropyLoss()(logits.float(), labels)
    assert abs(loss.item() - float(g["bf16__loss"])) <= 2e-3
    assert abs(loss.item() - float(g["fp32__loss"])) <= 1e-2
    loss.backward()
    api_grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    # --- fused path: loss + dlogits inside the pipeline
    la = m.forward_loss(x, mask, labels)
    assert abs(la[0].item() - float(g["bf16__loss"])) <= 2e-3
    m.backward()
    num = den = 0.0
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g[f"bf16__grad__{k}"]).cuda()
        for got in (p.grad, api_grads[k]):
            rel = (got - ref).norm() / (ref.norm() + 1e-12)
            assert rel <= 3e-2, (k, rel.item())
        num += (p.grad * ref).sum().item()
        den += (p.grad.norm() ** 2).item() ** 0.5 * 0 + 0
    flat_g = torch.cat([p.grad.flatten() for _, p in m.named_parameters()])
    flat_r = torch.cat([torch.from_numpy(g[f"bf16__grad__{k}"]).flatten() for k, _ in m.named_parameters()]).cuda()
    cos = torch.nn.functional.cosine_similarity(flat_g, flat_r, dim=0).item()
    assert cos >= 0.999, cos
    flat_r32 = torch.cat([torch.from_numpy(g[f"fp32__grad__{k}"]).flatten() for k, _ in m.named_parameters()]).cuda()
    assert torch.nn.functional.cosine_similarity(flat_g, flat_r32, dim=0).item() >= 0.995
    # --- return_all_tokens
    m.eval()
    with torch.no_grad():
        allt = m(x, mask, return_all_tokens=True)
    assert allt.shape == tuple(g["bf16__logits_all"].shape)
    assert np.abs(allt.float().cpu().numpy() - g["bf16__logits_all"]).max() <= 0.03


def test_long_nonsquare_vs_reference_golden():
    """More than 256 tokens (256 x 320 canvas, 16 x 20 window): the streaming attention kernels inside the
    whole model, against the reference's own bf16-autocast outputs (oracle/gen_golden_long.py)."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import vit_inputs
    from oracle.gen_golden_long import LONG, LONG_INPUTS
    from oracle.vit_ref import fill_by_name
    g = np.load(os.path.join(GOLDEN, "vit_long.npz"))
    m = pt_vit(**LONG)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=3))
    m = m.cuda().train()
    x, mask, labels = vit_inputs(LONG, *LONG_INPUTS)
    la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
    assert abs(la[0].item() - float(g["bf16__loss"])) <= 2e-3, (la[0].item(), float(g["bf16__loss"]))
    assert abs(la[0].item() - float(g["fp32__loss"])) <= 1e-2
    m.backward()
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g[f"bf16__grad__{k}"]).cuda()
        rel = (p.grad - ref).norm() / (ref.norm() + 1e-12)
        assert rel <= 3e-2, (k, rel.item())
    flat_g = torch.cat([p.grad.flatten() for _, p in m.named_parameters()])
    flat_r = torch.cat([torch.from_numpy(g[f"bf16__grad__{k}"]).flatten() for k, _ in m.named_parameters()]).cuda()
    assert torch.nn.functional.cosine_similarity(flat_g, flat_r, dim=0).item() >= 0.999
    m.eval()
    with torch.no_grad():
        lo = m(x.cuda(), mask.cuda())
    assert np.abs(lo.float().cpu().numpy() - g["bf16__logits"]).max() <= 0.03


def test_config5_geometry_vs_reference_golden():
    """BASELINE configs[4] geometry against the REFERENCE (oracle/gen_golden_c5.py -> tests/golden/vit_c5.npz): D = 1024,
    16 heads, 480 x 640 canvas = 1201 tokens (streaming attention kernels), the [4664, 16] relative-position table, 600
    masked patches, B = 1, depth 2.  bf16 product vs the reference's bf16-autocast run: loss, logits, every gradient tensor
    (small ones whole incl. the table gradient, big matrices on every 64th row), per-tensor norms, global direction; and
    the same outputs against the reference's fp32 run at the looser bf16-vs-fp32 bars."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import vit_inputs
    from oracle.gen_golden_c5 import C5, C5_INPUTS, sample
    from oracle.vit_ref import fill_by_name
    g = np.load(os.path.join(GOLDEN, "vit_c5.npz"))
    m = pt_vit(**C5)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=9))
    m = m.cuda().train()
    assert m.engine.T == 1201
    x, mask, labels = vit_inputs(C5, *C5_INPUTS)
    la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
    assert abs(la[0].item() - float(g["bf16__loss"])) <= 2e-3, (la[0].item(), float(g["bf16__loss"]))
    assert abs(la[0].item() - float(g["fp32__loss"])) <= 1e-2
    m.backward()
    num = den = dot = 0.0
    worst = (0.0, "")
    for k, p in m.named_parameters():
        got = sample(p.grad).float()
        ref = torch.from_numpy(g[f"bf16__grad__{k}"]).cuda()
        assert got.shape == ref.shape, k
        rel = ((got - ref).norm() / (ref.norm() + 1e-20)).item()
        worst = max(worst, (rel, k))
        # bf16 accumulation-order noise; the table gradient is a sum over 1201 x 1201 x B terms per bucket in fixed point
        assert rel <= (5e-2 if "relative_position_bias_table" in k else 3e-2), (k, rel)
        gn, rn = p.grad.double().norm().item(), float(g[f"bf16__gnorm__{k}"])
        assert abs(gn / rn - 1) <= 3e-2, (k, gn, rn)
        dot += float((got.double() * ref.double()).sum()); num += float((got.double() ** 2).sum()); den += float((ref.double() ** 2).sum())
        # direction against the reference's fp32 gradients (bf16-vs-fp32 bar)
        r32 = torch.from_numpy(g[f"fp32__grad__{k}"]).cuda()
        if r32.norm() > 0:
            cos32 = torch.nn.functional.cosine_similarity(got.flatten(), r32.flatten(), dim=0).item()
            assert cos32 >= 0.98, (k, cos32)
    print("config-5 geometry: worst per-tensor rel-L2 %.3e (%s), pooled cosine %.6f" % (worst[0], worst[1], dot / (num * den) ** 0.5))
    assert dot / (num * den) ** 0.5 >= 0.9995
    tab = m.rel_pos_bias.relative_position_bias_table.grad
    assert tuple(tab.shape) == (4664, 16) and torch.isfinite(tab).all()
    m.eval()
    with torch.no_grad():
        lo = m(x.cuda(), mask.cuda())
    # 98 304 logits against the reference's bf16-autocast run: the maximum is a tail event of bf16 rounding (one logit of
    # 98 304 sits 0.031 = four bf16 steps away with the round-5 attention kernels, 0.023 with the round-4 ones), so the bar
    # is the documented 0.04 for the maximum PLUS the body of the distribution (mean 0.0037, 99.9th percentile 0.0156 with
    # either kernel family; the reference's own bf16 run is 0.0052 / 0.028 from its fp32 run: tools/c5_logit_stats.py)
    d16 = np.abs(lo[:96].float().cpu().numpy() - g["bf16__logits_head"])
    assert d16.max() <= 0.04 and d16.mean() <= 0.0045 and np.quantile(d16, 0.999) <= 0.02, (d16.max(), d16.mean())
    assert np.abs(lo[:96].float().cpu().numpy() - g["fp32__logits_head"]).max() <= 0.06


def test_config5_geometry_20_training_steps_vs_reference():
    """BASELINE configs[4] geometry (1201 tokens: the long-window attention kernels are the only attention path it reaches), 20
    optimizer steps with the reference's optimizer recipe against the REFERENCE's curves (oracle/gen_golden_c5_train.py ->
    vit_c5_train20.npz): the bf16 engine follows the reference's bf16-autocast curve AND its fp32 curve to 2e-3 at every step
    (the two reference curves are within 4e-4 of each other on this run), gradient norms to 1 %; with the dS-storing attention
    backward (ViTEngine.attn_ds_workspace) the same."""
    import contextlib
    import io
    from mem_amd.modeling_pretrain import pt_vit
    from mem_amd.optim_factory import create_optimizer
    from mem_amd.utils import NativeScalerWithGradNormCount
    from oracle.gen_golden import vit_inputs
    from oracle.gen_golden_c5 import C5
    from oracle.vit_ref import fill_by_name
    g = np.load(os.path.join(GOLDEN, "vit_c5_train20.npz"))

    class A:
        opt = "adamw"; weight_decay = 0.05; lr = 5e-4; opt_eps = 1e-8; opt_betas = [0.9, 0.999]; momentum = 0.9
    for ds_ws in (False, True):
        m = pt_vit(**C5)
        m.load_state_dict(fill_by_name(m.state_dict(), seed=9))
        m = m.cuda().train()
        m.engine.attn_ds_workspace = ds_ws
        with contextlib.redirect_stdout(io.StringIO()):
            opt = create_optimizer(A(), m)
        scaler = NativeScalerWithGradNormCount()
        rec = []
        for it in range(20):
            for grp in opt.param_groups:
                grp["lr"] = g["lr"][it] * grp["lr_scale"]
                if grp["weight_decay"] > 0:
                    grp["weight_decay"] = g["wd"][it]
            x, mask, labels = vit_inputs(C5, 1, 3000 + it % 2, 600)
            la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
            m._fused_loss_pending = True
            gn = scaler(la, opt, clip_grad=30.0, parameters=m.parameters(), model=m)
            rec.append((la[0].item(), gn.item()))
        rec = np.array(rec)
        d16, d32 = np.abs(rec[:, 0] - g["bf16__loss"]), np.abs(rec[:, 0] - g["fp32__loss"])
        print("config-5 geometry, 20 steps (dS workspace %s): vs reference bf16 curve max %.2e, vs fp32 curve max %.2e (reference bf16 vs fp32: "
              "%.2e); grad-norm rel %.2e" % (ds_ws, d16.max(), d32.max(), np.abs(g["bf16__loss"] - g["fp32__loss"]).max(),
                                           np.abs(rec[:, 1] / g["bf16__gnorm"] - 1).max()))
        assert d16.max() <= 2e-3 and d32.max() <= 2e-3              # measured 4.2e-4 / 5.7e-4 (recomputing backward), 3.6e-4 / 5.0e-4 (dS workspace)
        assert np.abs(rec[:, 1] / g["bf16__gnorm"] - 1).max() <= 1e-2


def test_vit_large_480x640_step():
    """BASELINE configs[4] shapes: ViT-L/16 (D=1024, depth 24, 16 heads, layer scale 1e-5) on 480 x 640 2-bin
    voxels = 1201 tokens, 600 masked patches, B=2.  No CPU oracle at this size in seconds, so the checks are
    the size-independent ones: the loss at the reference's init (lm_head bias 0, weights ~ 0.02 trunc-normal)
    is ln 8192 plus half the logit variance (~0.06 for D=1024), every gradient is finite and non-zero, and a second identical step reproduces
    loss and gradients (nothing stale in the workspaces).  Numerics of this path are pinned at 321 tokens by
    test_long_nonsquare_vs_reference_golden and per kernel at 1201 tokens in test_kernels_gpu.py."""
    from mem_amd.modeling_pretrain import pt_vit
    cfg = dict(img_size=(480, 640), patch_size=(16, 16), in_chans=2, vocab_size=8192, embed_dim=1024, depth=24,
               num_heads=16, mlp_ratio=4, drop_path_rate=0.0, use_shared_rel_pos_bias=True, use_abs_pos_emb=False,
               init_values=1e-5)
    torch.manual_seed(0)
    m = pt_vit(**cfg).cuda().train()
    assert sum(p.numel() for p in m.parameters()) > 300e6
    g = torch.Generator().manual_seed(5)
    B, L = 2, 1200
    x = (torch.rand((B, 2, 480, 640), generator=g) * (torch.rand((B, 2, 480, 640), generator=g) < 0.3)).cuda()
    mask = torch.zeros((B, L), dtype=torch.bool)
    for b in range(B):
        mask[b, torch.randperm(L, generator=g)[:600]] = True
    labels = torch.randint(0, 8192, (int(mask.sum()),), generator=g).cuda()
    la = m.forward_loss(x, mask.cuda(), labels)
    assert 0.0 < la[0].item() - float(np.log(8192)) < 0.15, la[0].item()
    m.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    for k, gk in grads.items():
        assert torch.isfinite(gk).all(), k
        if "q_bias" not in k and "rel_pos" not in k:
            assert gk.abs().sum().item() > 0, k
    assert grads["rel_pos_bias.relative_position_bias_table"].shape == (59 * 79 + 3, 16)
    # a second, identical step reproduces the gradients (no stale accumulators across steps) ...
    la2 = m.forward_loss(x, mask.cuda(), labels)
    m.backward()
    assert la2[0].item() == la[0].item()
    for k, p in m.named_parameters():
        rel = (p.grad - grads[k]).norm() / (grads[k].norm() + 1e-20)
        assert rel <= 2e-2, (k, rel.item())          # table-gradient / bias atomics reorder fp32 sums


@pytest.mark.parametrize("tag", ["a", "b"])
def test_finetune_model_vs_reference_golden(tag):
    """f3: ft_vit on the fused engine (trunk in HIP, pooling / fc_norm / head in torch autograd) against the
    reference's bf16-autocast outputs: (a) per-block relative-position tables + mean pooling + layer scale (the
    finetuning default), (b) shared table + abs. position embedding + cls-token head, no layer scale."""
    from mem_amd.modeling_finetune import ft_vit
    from oracle.gen_golden_ft import FT_A, FT_B, ft_inputs
    from oracle.vit_ref import fill_by_name
    cfg = FT_A if tag == "a" else FT_B
    g = np.load(os.path.join(GOLDEN, "vit_ft.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "vit_ft_meta.json")))
    m = ft_vit(**cfg)
    assert list(m.state_dict().keys()) == meta[f"{tag}_state_keys"]
    m.load_state_dict(fill_by_name(m.state_dict(), seed=5))
    m = m.cuda().train()
    x, y = ft_inputs(cfg, 5, 31)
    lo = m(x.cuda())
    assert np.abs(lo.detach().float().cpu().numpy() - g[f"{tag}__logits"]).max() <= 0.03
    loss = torch.nn.CrossEntropyLoss()(lo.float(), y.cuda())
    assert abs(loss.item() - float(g[f"{tag}__loss"])) <= 3e-3
    loss.backward()
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g[f"{tag}__grad__{k}"]).cuda()
        rel = (p.grad - ref).norm() / (ref.norm() + 1e-12)
        assert rel <= 4e-2, (k, rel.item())
    flat_g = torch.cat([p.grad.flatten() for _, p in m.named_parameters()])
    flat_r = torch.cat([torch.from_numpy(g[f"{tag}__grad__{k}"]).flatten() for k, _ in m.named_parameters()]).cuda()
    assert torch.nn.functional.cosine_similarity(flat_g, flat_r, dim=0).item() >= 0.999
    m.eval()
    with torch.no_grad():
        lo2 = m(x.cuda())
    assert np.abs(lo2.float().cpu().numpy() - g[f"{tag}__logits"]).max() <= 0.03


def test_finetune_layer_decay_adamw_vs_torch():
    """Layer-wise lr decay: FlatAdamW's grouped update (memhip_adamw_groups) against torch.optim.AdamW with the same
    parameter groups, 3 steps with gradient clipping, on the finetuning model's parameters."""
    import contextlib
    import io
    from mem_amd import optim_factory as OF
    from mem_amd.modeling_finetune import ft_vit
    from oracle.gen_golden_ft import FT_A, ft_inputs
    from oracle.vit_ref import fill_by_name
    m = ft_vit(**FT_A)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=5))
    m = m.cuda().train()
    depth = FT_A["depth"]
    assigner = OF.LayerDecayValueAssigner(list(0.75 ** (depth + 1 - i) for i in range(depth + 2)))
    with contextlib.redirect_stdout(io.StringIO()):
        groups = OF.get_parameter_groups(m, 0.05, m.no_weight_decay(), assigner.get_layer_id, assigner.get_scale)
    opt = OF.FlatAdamW(m, groups, lr=1e-3)
    opt.max_norm = 1.0
    shadow = {k: p.detach().clone().requires_grad_(True) for k, p in m.named_parameters()}
    name_of = {id(p): n for n, p in m.named_parameters()}
    tgroups = [{"params": [shadow[name_of[id(p)]] for p in gr["params"]], "weight_decay": gr["weight_decay"],
                "lr": 1e-3 * gr["lr_scale"]} for gr in groups]
    topt = torch.optim.AdamW(tgroups, lr=1e-3, betas=(0.9, 0.95), eps=1e-8)
    x, y = ft_inputs(FT_A, 5, 31)
    for it in range(3):
        for gr in opt.param_groups:
            gr["lr"] = 1e-3 * gr["lr_scale"]
        loss = torch.nn.CrossEntropyLoss()(m(x.cuda()).float(), y.cuda())
        loss.backward()
        for k, p in m.named_parameters():
            shadow[k].grad = p.grad.detach().clone()
        torch.nn.utils.clip_grad_norm_(list(shadow.values()), 1.0)
        topt.step()
        m.engine.grad_norm()
        opt.step()
        for k, p in m.named_parameters():
            torch.testing.assert_close(p.detach(), shadow[k].detach(), rtol=2e-5, atol=2e-6, msg=lambda s_, k=k: f"{k}: {s_}")


def test_drop_path_masks_vs_oracle():
    """Stochastic depth with the keep masks fed in: product vs the CPU oracle under bf16 autocast."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import vit_inputs
    from oracle.vit_ref import RefViT, fill_by_name
    cfg = dict(TINY, drop_path_rate=0.3)
    m = pt_vit(**cfg)
    w = fill_by_name(m.state_dict(), seed=0)
    m.load_state_dict(w)
    m = m.cuda().train()
    o = RefViT(**cfg)
    o.load_state_dict(w)
    x, mask, labels = vit_inputs(cfg, 6, 5, 6)
    masks = (torch.rand(4, 6, generator=torch.Generator().manual_seed(1)) > 0.4).float()
    keep = [(masks[2 * i], masks[2 * i + 1]) for i in range(2)]
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lo = o(x, mask, keep=keep)
        lref = torch.nn.CrossEntropyLoss()(lo, labels)
    lref.backward()
    la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda(), drop_path_masks=masks.cuda())
    m.backward()
    assert abs(la[0].item() - lref.item()) <= 3e-3
    for (k, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        rel = (p.grad.cpu() - q.grad).norm() / (q.grad.norm() + 1e-12)
        assert rel <= 4e-2, (k, rel.item())


@pytest.mark.parametrize("fuse", [True, False])
def test_drop_path_work_skipping_equals_masked_execution(fuse):
    """Stochastic depth as work skipping (ViTEngine.dp_skip: every branch runs on its kept samples only, through compact
    batches and sample maps) against the masked execution of the same step (all samples computed, dropped ones multiplied
    by zero), at ViT-B width on the persistent 256-row GEMM kernels: same loss, same gradients (sums over fewer rows in
    another split order: fp32 round-off only), same residual stream.  Includes a branch that keeps everything, one that
    drops most, and a block without stochastic depth."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import BASE, vit_inputs
    from oracle.vit_ref import fill_by_name
    cfg = dict(BASE, in_chans=2, depth=4, drop_path_rate=0.3)
    B = 48
    x, mask, labels = vit_inputs(cfg, B, 21, 98)
    g = torch.Generator().manual_seed(5)
    masks = (torch.rand(8, B, generator=g) > 0.25).float()
    masks[2] = 1.0                                             # block 1, attention branch: nothing dropped this step
    masks[5] = (torch.rand(B, generator=g) > 0.8).float()      # block 2, MLP branch: most samples dropped
    res = {}
    for skip in (False, True):
        m = pt_vit(**cfg)
        m.load_state_dict(fill_by_name(m.state_dict(), seed=1))
        m = m.cuda().train()
        m.engine.dp_skip = skip
        m.engine.fuse_ln_branch = fuse
        m.engine.tail_rows = False                               # (the whole residual stream is compared below)
        la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda(), drop_path_masks=masks.cuda())
        m.backward()
        torch.cuda.synchronize()
        assert (m.engine.cur["plan"] is not None) == skip
        res[skip] = (la.clone(), m.engine.flat_g.clone(), m.engine.x[2 * 4][: B * 197].clone(), dict(m.engine.segs))
        del m
    (l0, g0, x0, segs), (l1, g1, x1, _) = res[False], res[True]
    assert torch.equal(x0, x1)                                 # forward: identical arithmetic per kept row
    assert abs(l0[0].item() - l1[0].item()) <= 1e-6 and l0[1].item() == l1[1].item()
    cos = torch.dot(g0, g1) / (g0.norm() * g1.norm())
    assert cos.item() >= 0.99999, cos.item()
    for name, (o, k) in segs.items():
        a, b = g0[o:o + k], g1[o:o + k]
        if float(a.norm()) > 0:
            rel = float((a - b).norm() / a.norm())
            assert rel <= 2e-3, (name, rel)                    # bf16 column-sum / split-order noise


@pytest.mark.parametrize("skip", [False, True])
def test_grouped_weight_gradient_launches_equal_single_launches(skip):
    """ViTEngine.wgrad_group: 0 = every weight gradient its own launch, 1 (default) = proj + qkv of a block as one grouped
    launch (memhip_gemm_bf16_tn_group), 2 = fc2 + fc1 too.  Same step, same gradients: the weight gradients differ by the
    row-slice boundaries of their split only (fp32 round-off), the layer-scale gradients are linear functions of them.  With and without the work-skipping plan of stochastic depth (different row counts for
    the two branches of a block)."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import BASE, vit_inputs
    from oracle.vit_ref import fill_by_name
    cfg = dict(BASE, in_chans=2, depth=3, drop_path_rate=0.3 if skip else 0.0)
    B = 40
    x, mask, labels = vit_inputs(cfg, B, 23, 98)
    g = torch.Generator().manual_seed(7)
    masks = (torch.rand(6, B, generator=g) > 0.3).float() if skip else None
    res = {}
    for level in (0, 1, 2):
        m = pt_vit(**cfg)
        m.load_state_dict(fill_by_name(m.state_dict(), seed=2))
        m = m.cuda().train()
        m.engine.wgrad_group = level
        la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda(), drop_path_masks=masks.cuda() if skip else None)
        m.backward()
        torch.cuda.synchronize()
        assert (m.engine.cur["plan"] is not None) == skip
        res[level] = (la[0].item(), m.engine.flat_g.clone(), dict(m.engine.segs))
        del m
    l0, g0, segs = res[0]
    for level in (1, 2):
        l1, g1, _ = res[level]
        assert l1 == l0
        for name, (o, k) in segs.items():
            a, b = g0[o:o + k], g1[o:o + k]
            # (regrouped weight gradients: another split of the same fp32 sum; the rest: the same kernels, of which the
            # column-sum / LayerNorm / table gradients add with fp32 atomics in an order that varies from run to run)
            assert float((a - b).norm()) <= 1e-5 * float(a.norm()) + 1e-12, (level, name)


@pytest.mark.parametrize("skip", [False, True])
def test_last_block_tail_rows_equal_full_execution(skip):
    """Dead-row elimination in the last block (ViTEngine.tail_rows: its MLP branch runs only on the rows that reach the head,
    mem/modeling_pretrain.py:119-126) against the full execution of the same step, with stochastic depth on the last block's
    MLP branch, in the masked and in the work-skipping mode: the rows that reach the head are bit-identical (so are logits
    and loss), gradients equal up to the order of fp32 sums."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import BASE, vit_inputs
    from oracle.vit_ref import fill_by_name
    cfg = dict(BASE, in_chans=2, depth=3, drop_path_rate=0.3)
    B = 40
    x, mask, labels = vit_inputs(cfg, B, 23, 98)
    g = torch.Generator().manual_seed(9)
    masks = (torch.rand(6, B, generator=g) > 0.25).float()
    res = {}
    for tail in (False, True):
        m = pt_vit(**cfg)
        m.load_state_dict(fill_by_name(m.state_dict(), seed=1))
        m = m.cuda().train()
        m.engine.dp_skip = skip
        m.engine.tail_rows = tail
        la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda(), drop_path_masks=masks.cuda())
        eng = m.engine
        assert (eng.cur["tail"] is not None) == tail
        rows = eng.cur["rows"].long()
        xrows = eng.x_tail[: rows.numel()].clone() if tail else eng.x[2 * 3].index_select(0, rows)
        m.backward()
        torch.cuda.synchronize()
        res[tail] = (la.clone(), eng.flat_g.clone(), xrows, dict(eng.segs))
        del m
    (l0, g0, x0, segs), (l1, g1, x1, _) = res[False], res[True]
    assert torch.equal(x0, x1)                                   # the rows that reach the head: same arithmetic
    assert l0[0].item() == l1[0].item() and l0[1].item() == l1[1].item()
    cos = torch.dot(g0, g1) / (g0.norm() * g1.norm())
    assert cos.item() >= 0.99999, cos.item()
    for name, (o, k) in segs.items():
        a, b = g0[o:o + k], g1[o:o + k]
        if float(a.norm()) > 0:
            rel = float((a - b).norm() / a.norm())
            assert rel <= 2e-3, (name, rel)


@pytest.mark.parametrize("C", [3, 2])
def test_vit_base_vs_reference_golden(C):
    """ViT-B/16 224^2 (BASELINE shapes), B=2: loss, sampled logits and per-parameter gradient norms."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import BASE, vit_inputs
    from oracle.vit_ref import fill_by_name
    g = np.load(os.path.join(GOLDEN, f"vit_base_c{C}.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "vit_meta.json")))
    cfg = dict(BASE, in_chans=C)
    m = pt_vit(**cfg)
    assert sum(p.numel() for p in m.parameters()) == meta[f"base_c{C}_nparams"]
    m.load_state_dict(fill_by_name(m.state_dict(), seed=1))
    m = m.cuda().train()
    x, mask, labels = vit_inputs(cfg, 2, 77, 98)
    la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
    assert abs(la[0].item() - float(g["bf16__loss"])) <= 5e-3, (la[0].item(), float(g["bf16__loss"]))
    assert abs(la[0].item() - float(g["fp32__loss"])) <= 2e-2
    m.eval()
    with torch.no_grad():
        lo = m(x.cuda(), mask.cuda())
    m.train()
    samp = lo.float().cpu()[::7, ::97].numpy()
    # 2 380 sampled logits (|logit| <= 2.6: one bf16 ulp = 0.0156).  Measured on MI355X: max 0.0195 / 0.0205, mean 0.0044 / 0.0043,
    # rel-L2 0.0065 / 0.0068 (C = 2 / 3) -- and against the reference's fp32 run our bf16 logits deviate as much as the reference's
    # own bf16-autocast logits do (mean 0.0060 vs 0.0059, max 0.024 vs 0.025): the bars below hold both statements
    refl, ref32 = g["bf16__logits_sample"], g["fp32__logits_sample"]
    dl = np.abs(samp - refl)
    assert dl.max() <= 0.03 and dl.mean() <= 0.007, (dl.max(), dl.mean())
    assert np.linalg.norm(samp - refl) / np.linalg.norm(refl) <= 0.01
    assert np.abs(samp - ref32).mean() <= 1.25 * np.abs(refl - ref32).mean(), (np.abs(samp - ref32).mean(), np.abs(refl - ref32).mean())
    m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
    m.backward()
    names = meta[f"base_c{C}_param_names"]
    gn = np.array([dict(m.named_parameters())[k].grad.norm().item() for k in names])
    ref = g["bf16__gradnorms"]
    big = ref > 1e-6
    dev = np.abs(gn / np.maximum(ref, 1e-30) - 1) * big
    order = np.argsort(-dev)[:3]
    print("ViT-B C=%d per-tensor gradient-norm deviation from the reference (bf16): worst " % C
          + ", ".join("%s %.4f" % (names[i], dev[i]) for i in order))
    assert np.abs(gn[big] / ref[big] - 1).max() <= 0.015, np.abs(gn[big] / ref[big] - 1).max()     # measured: 0.0047 (C=2), 0.0018 (C=3)
    tot = np.sqrt((gn ** 2).sum()) / np.sqrt((ref ** 2).sum())
    assert abs(tot - 1) <= 0.01, tot
    # direction at ViT-B size: full gradient tensors of a cross-section of parameters (whole small tensors, strided
    # rows of the big matrices) against the reference's bf16-autocast gradients: rel-L2 <= 3e-2 each, cosine >= 0.999
    from oracle.gen_golden import BASE_GRAD_SAMPLES
    pd = dict(m.named_parameters())
    got_all, ref_all = [], []
    for k, st in BASE_GRAD_SAMPLES:
        r = torch.from_numpy(g[f"bf16__grad__{k}"]).cuda()
        q = pd[k].grad[::st]
        rel = ((q - r).norm() / (r.norm() + 1e-12)).item()
        cos = torch.nn.functional.cosine_similarity(q.flatten(), r.flatten(), dim=0).item()
        assert rel <= 3e-2 and cos >= 0.999, (k, rel, cos)
        got_all.append(q.flatten()); ref_all.append(r.flatten())
        r32 = torch.from_numpy(g[f"fp32__grad__{k}"]).cuda()
        assert torch.nn.functional.cosine_similarity(q.flatten(), r32.flatten(), dim=0).item() >= 0.995, k
    cos = torch.nn.functional.cosine_similarity(torch.cat(got_all), torch.cat(ref_all), dim=0).item()
    assert cos >= 0.9995, cos


def test_batch_is_token_weighted_mean_of_samples():
    """Size-independent property of the masked-token loss (engine_for_pretraining.py:152: mean CE over the batch's M
    masked tokens): the loss / gradient of a batch equals the M_b-weighted mean of the per-sample losses / gradients.
    Exercises the ragged row gather, the per-sample mask-token blend and every batched reduction (bias, LayerNorm,
    layer-scale, table gradients) against single-sample runs of the same engine."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import vit_inputs
    from oracle.vit_ref import fill_by_name
    m = pt_vit(**TINY)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=0))
    m = m.cuda().train()
    B = 5
    x, mask, labels = vit_inputs(TINY, B, 41, 7)
    counts = mask.sum(1).tolist()
    la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
    m.backward()
    g_all = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    loss_all = la[0].item()
    acc = {k: torch.zeros_like(v) for k, v in g_all.items()}
    loss_acc, off = 0.0, 0
    for b in range(B):
        lb = labels[off:off + counts[b]]
        off += counts[b]
        l1 = m.forward_loss(x[b:b + 1].cuda(), mask[b:b + 1].cuda(), lb.cuda())
        m.backward()
        w = counts[b] / sum(counts)
        loss_acc += w * l1[0].item()
        for k, p in m.named_parameters():
            acc[k] += w * p.grad
    assert abs(loss_all - loss_acc) <= 2e-4, (loss_all, loss_acc)
    for k in g_all:
        rel = (g_all[k] - acc[k]).norm() / (acc[k].norm() + 1e-12)
        assert rel <= 2e-2, (k, rel.item())           # bf16 dlogits scale differently (1/M vs 1/M_b): rounding noise only


def test_vit_base_full_batch_is_weighted_mean_of_halves():
    """BASELINE configs[1] size (ViT-B/16, C=2, B=256, ~98 masked patches per sample): the batch loss / gradients equal
    the masked-token-weighted mean of the two half batches -- exercises the row-split GEMM launches, the 13-samples-
    per-workgroup attention grids and every batched reduction at the size the benchmark runs."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import BASE, vit_inputs
    from oracle.vit_ref import fill_by_name
    cfg = dict(BASE, in_chans=2)
    m = pt_vit(**cfg)
    m.load_state_dict(fill_by_name(m.state_dict(), seed=1))
    m = m.cuda().train()
    B = 256
    x, mask, labels = vit_inputs(cfg, B, 5, 98)
    counts = mask.sum(1)
    xs, ms, ls = x.cuda(), mask.cuda(), labels.cuda()
    la = m.forward_loss(xs, ms, ls)
    m.backward()
    loss_all = la[0].item()
    assert abs(loss_all - float(np.log(8192))) < 1.5          # recipe-filled weights: ln V plus half the logit variance
    names = ["blocks.0.attn.qkv.weight", "blocks.11.mlp.fc2.weight", "blocks.5.norm1.weight", "blocks.7.attn.q_bias",
             "rel_pos_bias.relative_position_bias_table", "patch_embed.proj.weight", "lm_head.bias", "blocks.3.gamma_1",
             "cls_token", "mask_token"]
    P = dict(m.named_parameters())
    g_all = {k: P[k].grad.detach().clone() for k in names}
    for k, p in P.items():
        assert torch.isfinite(p.grad).all(), k
    acc = {k: torch.zeros_like(v) for k, v in g_all.items()}
    loss_acc, off = 0.0, 0
    for lo, hi in ((0, 128), (128, 256)):
        nb = int(counts[lo:hi].sum())
        l1 = m.forward_loss(xs[lo:hi], ms[lo:hi], ls[off:off + nb])
        m.backward()
        off += nb
        w = nb / int(counts.sum())
        loss_acc += w * l1[0].item()
        for k in names:
            acc[k] += w * P[k].grad
    assert abs(loss_all - loss_acc) <= 2e-4, (loss_all, loss_acc)
    for k in names:
        rel = (g_all[k] - acc[k]).norm() / (acc[k].norm() + 1e-12)
        assert rel <= 2e-2, (k, rel.item())


def test_odd_depth_repeated_backward_vs_oracle():
    """Depth 3 (odd): the proj-bias / v_bias ping-pong scratch of the backward must start clean on EVERY backward
    (round-1 defect: from the second backward on, the last block's attn.proj.bias / attn.v_bias gradients carried
    block 0's column sums of the previous step).  Checker: the oracle model on the host CPU under bf16 autocast."""
    from mem_amd.modeling_pretrain import pt_vit
    from oracle.gen_golden import vit_inputs
    from oracle.vit_ref import RefViT, fill_by_name
    cfg = dict(TINY, depth=3)
    m = pt_vit(**cfg)
    sd = fill_by_name(m.state_dict(), seed=5)
    m.load_state_dict(sd)
    m = m.cuda().train()
    o = RefViT(**cfg)
    o.load_state_dict(sd)
    o.train()
    for step in range(3):                                   # different inputs each time, no optimizer step
        x, mask, labels = vit_inputs(cfg, 4, 900 + step, 6)
        for p in o.parameters():
            p.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lo = o(x, mask)
        loss_o = torch.nn.CrossEntropyLoss()(lo.float(), labels)
        loss_o.backward()
        la = m.forward_loss(x.cuda(), mask.cuda(), labels.cuda())
        m.backward()
        assert abs(la[0].item() - loss_o.item()) <= 2e-3
        og = dict(o.named_parameters())
        for k, p in m.named_parameters():
            ref = og[k].grad.cuda()
            rel = (p.grad - ref).norm() / (ref.norm() + 1e-12)
            assert rel <= 3e-2, (step, k, rel.item())
