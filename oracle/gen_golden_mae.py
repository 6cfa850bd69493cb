"""Generate tests/golden/mae_tiny.npz from the REFERENCE MaskedAutoencoderViT (build container only).

The reference class is imported as it is; timm's PatchEmbed / Block (un-vendored) are stood in by the restatements of
oracle/mae_ref.py.  torch.rand inside random_masking is patched for the duration of one forward so that the noise is a
stored input.  Asserted bit-exact: same-seed initial weights, loss, pred, mask and every gradient of the reference ==
oracle RefMAE; both loss modes.     python -m oracle.gen_golden_mae"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refimport as R            # noqa: E402
from oracle import mae_ref as M               # noqa: E402


def main():
    if not R.install():
        sys.exit("no /root/reference here")
    torch.set_num_threads(1)
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.PatchEmbed, vt.Block = M.PatchEmbed, M.Block
    sys.modules["timm.models.vision_transformer"] = vt
    import contextlib, io
    import modeling_mae as RM                                   # the reference module
    res = {}
    for mode in (True, False):
        cfg = dict(M.TINY_MAE, LOSS_ONLY_MASKED_MAE=mode)
        with contextlib.redirect_stdout(io.StringIO()):
            torch.manual_seed(3); ref = RM.MaskedAutoencoderViT(**cfg)
        torch.manual_seed(3); ora = M.RefMAE(**cfg)
        for (k, a), (k2, b) in zip(ref.state_dict().items(), ora.state_dict().items()):
            assert k == k2 and torch.equal(a, b), k
        imgs, noise = M.mae_inputs(cfg, 4, 21)
        real_rand = torch.rand
        torch.rand = lambda *a, **k: noise.clone()              # random_masking's draw (modeling_mae.py:213)
        try:
            loss_r, img_r, mask_r = ref(imgs)
        finally:
            torch.rand = real_rand
        loss_o, pred_o, mask_o = ora(imgs, noise)
        assert torch.equal(loss_r, loss_o) and torch.equal(mask_r, mask_o) and torch.equal(ref.patchify(img_r), pred_o)
        loss_r.backward(); loss_o.backward()
        tag = "masked" if mode else "all"
        for (k, p), (_, q) in zip(ref.named_parameters(), ora.named_parameters()):
            if p.requires_grad:
                assert torch.equal(p.grad, q.grad), k
                if mode or k in ("cls_token", "mask_token", "decoder_pred.bias", "blocks.0.attn.qkv.weight", "patch_embed.proj.bias"):
                    res[f"{tag}__grad__{k}"] = p.grad.numpy()      # "all" mode: a cross-section only (fixture size)
        res[f"{tag}__loss"] = loss_r.detach().numpy(); res[f"{tag}__pred"] = pred_o.detach().numpy(); res[f"{tag}__mask"] = mask_r.numpy()
    res["keys"] = np.array(list(ref.state_dict().keys()))
    out = os.path.join(ROOT, "tests", "golden", "mae_tiny.npz")
    np.savez_compressed(out, **res)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB; loss masked/all:", float(res["masked__loss"]), float(res["all__loss"]))
    gen_bf16(RM)


SMALL = 1 << 16


def sample(t):
    """What the bf16 fixture keeps of a gradient tensor: small tensors whole, big matrices every 32nd row."""
    return t if t.numel() <= SMALL else t[::32]


BASE_MAE_B = 2


def gen_bf16(RM):
    """The reference under bf16 autocast (mem/engine_for_pretraining.py:141-149 runs the MAE model under autocast like
    pt_vit): tiny config (every gradient whole) and the ViT-B factory at B = 2 (loss, a block of the prediction, every
    gradient's norm, small gradients whole, big ones on every 32nd row) -> tests/golden/mae_bf16.npz.  The oracle
    restatement is asserted bit-equal to the reference in the same process first."""
    import contextlib, io
    from functools import partial
    import torch.nn as nn
    res = {}
    base = dict(img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, decoder_embed_dim=512,
                decoder_depth=8, decoder_num_heads=16, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6))
    for tag, cfg0, B, seed, threads in (("tiny", M.TINY_MAE, 4, 21, 1), ("base", base, BASE_MAE_B, 33, 8)):
        torch.set_num_threads(threads)
        cfg = dict(cfg0, LOSS_ONLY_MASKED_MAE=True)
        with contextlib.redirect_stdout(io.StringIO()):
            torch.manual_seed(3); ref = RM.MaskedAutoencoderViT(**cfg)
        torch.manual_seed(3); ora = M.RefMAE(**cfg)
        imgs, noise = M.mae_inputs(cfg, B, seed)
        real_rand = torch.rand
        torch.rand = lambda *a, **k: noise.clone()
        try:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                loss_r, img_r, mask_r = ref(imgs)
        finally:
            torch.rand = real_rand
        with torch.autocast("cpu", dtype=torch.bfloat16):
            loss_o, pred_o, mask_o = ora(imgs, noise)
        assert torch.equal(loss_r, loss_o) and torch.equal(mask_r, mask_o) and torch.equal(ref.patchify(img_r).float(), pred_o.float())
        loss_r.backward(); loss_o.backward()
        for (k, p), (_, q) in zip(ref.named_parameters(), ora.named_parameters()):
            if p.requires_grad:
                assert torch.equal(p.grad, q.grad), k
                res[f"{tag}__gnorm__{k}"] = np.float64(p.grad.double().norm().item())
                res[f"{tag}__grad__{k}"] = sample(p.grad).numpy()
        res[f"{tag}__loss"] = loss_r.detach().float().numpy()
        res[f"{tag}__pred_head"] = pred_o.detach().float()[:, :24].numpy()
        res[f"{tag}__mask"] = mask_r.numpy()
    out = os.path.join(ROOT, "tests", "golden", "mae_bf16.npz")
    np.savez_compressed(out, **res)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB; bf16 loss tiny/base:", float(res["tiny__loss"]), float(res["base__loss"]))


if __name__ == "__main__":
    main()
