"""-m gpu: the MAE variant (mem_amd/modeling_mae.py -> csrc/fp32_path.hip: fp32 GEMMs, generic attention with 64- and
32-wide heads, token gather / mask-token un-shuffle, per-patch MSE loss) against the REFERENCE's MaskedAutoencoderViT
(tests/golden/mae_tiny.npz, oracle/gen_golden_mae.py: fp32 CPU, timm Block / PatchEmbed restated): same-seed weights
equal, loss 2e-6 rel, pred 2e-5, mask equal, every gradient rel-L2 <= 2e-5; both loss modes; a short training run
through engine_for_pretraining.train_one_epoch(MAE=True); the ViT-B MAE factory shapes.
Reference: mem/modeling_mae.py:101-313, mem/run_mem_pretraining.py:231-232,275-276, mem/engine_for_pretraining.py:141-149."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _tiny(mode, seed=3):
    from mem_amd.modeling_mae import MaskedAutoencoderViT
    from oracle.mae_ref import TINY_MAE
    with contextlib.redirect_stdout(io.StringIO()):
        torch.manual_seed(seed)
        m = MaskedAutoencoderViT(**dict(TINY_MAE, LOSS_ONLY_MASKED_MAE=mode))
    return m


@pytest.mark.parametrize("mode", [True, False])
def test_mae_tiny_vs_reference_golden(mode):
    from oracle.mae_ref import TINY_MAE, RefMAE, mae_inputs
    g = np.load(os.path.join(GOLDEN, "mae_tiny.npz"))
    tag = "masked" if mode else "all"
    m = _tiny(mode)
    assert list(m.state_dict().keys()) == list(g["keys"])
    torch.manual_seed(3)
    ora = RefMAE(**dict(TINY_MAE, LOSS_ONLY_MASKED_MAE=mode))          # same seed -> same initial weights as the reference
    for (k, a), (_, b) in zip(m.state_dict().items(), ora.state_dict().items()):
        assert torch.equal(a, b), k
    m = m.cuda().train()
    imgs, noise = mae_inputs(TINY_MAE, 4, 21)
    loss, img, mask = m(imgs.cuda(), noise=noise.cuda())
    assert abs(loss.item() / float(g[f"{tag}__loss"]) - 1) <= 2e-6, (loss.item(), float(g[f"{tag}__loss"]))
    assert np.array_equal(mask.cpu().numpy(), g[f"{tag}__mask"])
    assert np.abs(m.patchify(img).cpu().numpy() - g[f"{tag}__pred"]).max() <= 2e-5
    m.forward_loss(imgs.cuda(), noise=noise.cuda())
    m.backward()
    checked = 0
    for k, p in m.named_parameters():
        key = f"{tag}__grad__{k}"
        if key not in g.files:
            continue
        ref = torch.from_numpy(g[key]).cuda()
        rel = ((p.grad - ref).norm() / (ref.norm() + 1e-20)).item()
        assert rel <= 2e-5, (k, rel)
        checked += 1
    assert checked >= (40 if mode else 5)
    assert m.pos_embed.grad is None and not m.pos_embed.requires_grad


def test_mae_training_loop_and_factory(tmp_path):
    """train_one_epoch(MAE=True) on a small model: loss decreases; the base factory has the reference's shapes."""
    from mem_amd import engine_for_pretraining as E
    from mem_amd.modeling_mae import mae_vit_base_patch16_dec512d8b
    from mem_amd.optim_factory import create_optimizer
    from mem_amd.utils import NativeScalerWithGradNormCount
    m = _tiny(True).cuda()

    class A:
        opt = "adamw"; weight_decay = 0.05; lr = 1e-3; opt_eps = 1e-8; opt_betas = [0.9, 0.999]; momentum = 0.9
    with contextlib.redirect_stdout(io.StringIO()):
        opt = create_optimizer(A(), m)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(8, 3, 64, 64, generator=g)
    batches = [((x, x, torch.zeros(8, 4, 4, dtype=torch.int64)), 0)] * 40
    lr = np.full(40, 1e-3)
    with contextlib.redirect_stdout(io.StringIO()):
        stats = E.train_one_epoch(m, None, batches[:20], opt, torch.device("cuda"), 0, NativeScalerWithGradNormCount(), 1.0,
                                  lr_schedule_values=lr, start_steps=0, MAE=True)
        stats2 = E.train_one_epoch(m, None, batches[20:], opt, torch.device("cuda"), 1, NativeScalerWithGradNormCount(), 1.0,
                                   lr_schedule_values=lr, start_steps=20, MAE=True)
        ev = E.evaluate(batches[:2], m, None, torch.device("cuda"), None, MAE=True)
    assert np.isfinite(stats["loss"]) and stats2["loss"] < stats["loss"] and stats["mlm_acc"] == 0
    assert np.isfinite(ev["loss"])
    with contextlib.redirect_stdout(io.StringIO()):
        b = mae_vit_base_patch16_dec512d8b(norm_pix_loss=0, LOSS_ONLY_MASKED_MAE=True)
    sd = b.state_dict()
    assert sd["pos_embed"].shape == (1, 197, 768) and sd["decoder_pos_embed"].shape == (1, 197, 512)
    assert sd["decoder_blocks.7.attn.qkv.weight"].shape == (1536, 512) and sd["decoder_pred.weight"].shape == (768, 512)
    assert len(b.blocks) == 12 and b.decoder_blocks[0].attn.num_heads == 16 and b.norm.eps == 1e-6


def test_cli_mae_flag(tmp_path):
    """`--mae 1` through the entrypoint (run_mem_pretraining.py:231-232,275-276): ViT-B MAE factory on 112^2 synthetic
    event frames, one short epoch + eval, checkpoint and log written."""
    import json
    from mem_amd.run_mem_pretraining import get_args, main
    out = tmp_path / "run"
    out.mkdir()
    args = get_args(["--expweek", "t", "--mae", "1", "--data_path", "synthetic", "--input_H", "112", "--input_W", "112",
                     "--batch_size", "4", "--epochs", "1", "--warmup_epochs", "0", "--synthetic_samples", "8",
                     "--num_workers", "0", "--output_dir", str(out), "--color_jitter", "0", "--rand_aug", "0",
                     "--slice_max_evs", "5000", "--clip_grad", "3.0"])
    main(args)
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert len(log) == 1 and np.isfinite(log[0]["train_loss"]) and log[0]["train_mlm_acc"] == 0
    sd = torch.load(next(p for p in out.iterdir() if p.name.startswith("checkpoint-")), map_location="cpu", weights_only=False)
    assert "decoder_blocks.7.mlp.fc2.weight" in sd["model"] and "pos_embed" in sd["model"]
