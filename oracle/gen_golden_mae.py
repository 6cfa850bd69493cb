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


if __name__ == "__main__":
    main()
