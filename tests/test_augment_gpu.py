"""-m gpu: the batched on-GPU augmentation chain (mem_amd/augment.py -> csrc/augment.hip, raster.hip) against
  * the fixtures written from the REFERENCE's EventRandAugment (tests/golden/randaug.npz): uint8, held to equality
    except where a float32 product differs by an ulp from torch's CPU kernels right at a rounding boundary
    (stated bar: <= 0.05 % of the pixels, each by exactly one level);
  * the oracle chain (oracle/aug_chain.py) on seeded synthetic streams with the geometry of configs/ncaltech.conf
    (per-sample canvases inferred from the data), nimagenet.conf (scaled events, 256 x 341 canvas, RandomCrop) and DSEC:
    canvas sizes and voxel counts bit-exact, float stages <= 2e-6, uint8 stage as above.
Reference: mem/datasets.py:611-660,26-82; mem/transforms.py:292-484."""
import contextlib
import io
import os
import random
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import aug_chain as AC
from oracle import aug_t as A
from oracle.gen_golden_aug import event_like_u8

pytestmark = pytest.mark.gpu

U8_FRAC = 5e-4          # pixels that may differ (by one level) after a resampled / blended uint8 stage


def _u8_close(got, want, what, one_level=True):
    d = (got.astype(np.int16) - want.astype(np.int16))
    if one_level:                 # a single op; behind Posterize / Solarize / Equalize a one-level difference is amplified
        assert np.abs(d).max() <= 1, (what, np.abs(d).max())
    frac = float((d != 0).mean())
    assert frac <= U8_FRAC, (what, frac)
    return frac


def _run_ops(img_u8, ops, mags):
    from mem_amd import augment as AG
    from mem_amd._lib import check, lib, ptr, stream_ptr
    a = img_u8[None].cuda().contiguous()
    b = torch.empty_like(a)
    for o, m in zip(ops, mags):
        rec = AG.randaug_record(A.OPS[int(o)], float(m)).reshape(1)
        rd = torch.from_numpy(rec.view(np.uint8).copy()).cuda()
        check(lib.memhip_rand_augment_u8(ptr(a), ptr(b), ptr(rd), 1, a.shape[2], a.shape[3], stream_ptr()), "ra")
        a, b = b, a
    return a[0].cpu().numpy()


def test_randaug_ops_vs_reference_golden():
    g = np.load(os.path.join(GOLDEN, "randaug.npz"))
    exact = 0
    for s in g["seeds"]:
        got = _run_ops(event_like_u8(int(s)), g[f"s{s}__ops"], g[f"s{s}__mags"])
        frac = _u8_close(got, g[f"s{s}__out"], f"seed {s} ops {g[f's{s}__ops']}", one_level=False)
        exact += frac == 0.0
    print("bit-exact cases: %d / %d" % (exact, len(g["seeds"])))
    assert exact >= len(g["seeds"]) // 2


@pytest.mark.parametrize("name,mag", [("ShearX", 0.3), ("ShearY", -0.17), ("TranslateX", 77.2), ("TranslateY", -101.0),
                                      ("Rotate", 19.0), ("Rotate", -30.0), ("Brightness", 0.9), ("Brightness", -0.6),
                                      ("Color", 0.45), ("Color", -0.9), ("Contrast", 0.6), ("Contrast", -0.33),
                                      ("Sharpness", 0.9), ("Sharpness", -0.9), ("Posterize", 4), ("Posterize", 8),
                                      ("Solarize", 127.5), ("Solarize", 0.0), ("AutoContrast", 0.0), ("Equalize", 0.0),
                                      ("Identity", 0.0)])
def test_each_op_vs_oracle(name, mag):
    for s in (1, 3, 7, 14):                       # incl. live middle channel and a constant channel
        img = event_like_u8(s)
        got = _run_ops(img, [A.OPS.index(name)], [mag])
        want = A.apply_op(img, name, mag).numpy()
        if name in ("Posterize", "Solarize", "AutoContrast", "Equalize", "Identity", "Brightness"):
            assert np.array_equal(got, want), (name, s)
        else:
            _u8_close(got, want, (name, mag, s))
    small = torch.randint(0, 256, (3, 37, 53), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    got = _run_ops(small, [A.OPS.index(name)], [mag])   # odd, non-square sizes
    _u8_close(got, A.apply_op(small, name, mag).numpy(), (name, "37x53"))


@pytest.mark.parametrize("h,w,oh,ow", [(180, 240, 224, 224), (173, 201, 224, 224), (100, 120, 224, 224), (224, 224, 224, 224),
                                       (440, 640, 224, 224), (120, 100, 224, 224), (300, 224, 224, 224),
                                       # downscales beyond 3.5x (more than 8 taps per axis): the CLI default --input_H/W 128
                                       # on a DSEC canvas (5x: 11 taps), 480 x 640 -> 64 x 64 (10x: 21 taps), 6.9x / 1.1x mixed
                                       (440, 640, 128, 128), (480, 640, 64, 64), (440, 141, 64, 128)])
def test_resize_antialias_vs_oracle(h, w, oh, ow):
    from mem_amd._lib import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(h * 1000 + w)
    B = 3
    img = torch.randint(0, 256, (B, 3, h, w), dtype=torch.uint8, generator=g)
    img[:, :, :, ::3] = 0
    out = torch.empty((B, 3, oh, ow), device="cuda")
    d = img.cuda()
    check(lib.memhip_resample_to_f32(ptr(d), None, 3 * h * w, h, w, 0, None, B, oh, ow, ptr(out), stream_ptr()), "resize")
    want = torch.stack([A.resize_bilinear_aa(img[b].float().div(255), (oh, ow)) for b in range(B)])
    assert (out.cpu() - want).abs().max().item() <= 2e-6


def test_crop_and_pad_vs_oracle():
    from mem_amd._lib import check, lib, ptr, stream_ptr
    for (h, w), (i, j) in (((256, 341), (17, 100)), ((256, 341), (32, 117)), ((200, 150), (60, 11)), ((224, 224), (0, 0))):
        img = torch.randint(0, 256, (2, 3, h, w), dtype=torch.uint8, generator=torch.Generator().manual_seed(h + i))
        ph, pw = A.padded_size(h, w, 224, 224)
        i, j = min(i, ph - 224), min(j, pw - 224)
        offs = torch.tensor([[i, j], [0, 0]], dtype=torch.int32).cuda()
        out = torch.empty((2, 3, 224, 224), device="cuda")
        imd = img.cuda()
        check(lib.memhip_resample_to_f32(ptr(imd), None, 3 * h * w, h, w, 1, ptr(offs), 2, 224, 224, ptr(out), stream_ptr()), "crop")
        assert torch.equal(out[0].cpu(), A.random_crop(img[0].float().div(255), 224, 224, i, j))
        assert torch.equal(out[1].cpu(), A.random_crop(img[1].float().div(255), 224, 224, 0, 0))


def test_color_jitter_vs_oracle():
    from mem_amd import augment as AG
    from mem_amd._lib import check, lib, ptr, stream_ptr
    torch.manual_seed(3)
    B = 8
    x = torch.rand(B, 3, 64, 48) * (torch.rand(B, 3, 64, 48) < 0.4)
    draws = [A.color_jitter_draw(0.4, 0.4) for _ in range(B - 2)] + [([2, 0, 1, 3], None, 1.3), ([0, 1, 2, 3], 0.7, None)]
    rec = np.stack([AG.jitter_record(*d) for d in draws])
    out = torch.empty((B, 3, 64, 48), device="cuda")
    xd, rd = x.cuda(), torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).cuda()
    check(lib.memhip_color_jitter(ptr(xd), 0, B, 64, 48, ptr(rd), ptr(out), 3, stream_ptr()), "cj")
    want = torch.stack([A.color_jitter(x[b], *draws[b]) for b in range(B)])
    assert (out.cpu() - want).abs().max().item() <= 1.2e-7
    # u8 input + channel drop (ToFloat32 and the 2-bin view fused), no jitter
    u = torch.randint(0, 256, (2, 3, 64, 48), dtype=torch.uint8)
    o2 = torch.empty((2, 2, 64, 48), device="cuda")
    ud = u.cuda()
    check(lib.memhip_color_jitter(ptr(ud), 1, 2, 64, 48, None, ptr(o2), 2, stream_ptr()), "cj")
    assert torch.equal(o2.cpu(), (u.float() / 255)[:, 0::2])


def _args(**kw):
    a = dict(data_path="x/ncaltech101/", input_H=224, input_W=224, slice_max_evs=30000, max_random_shift_evs=8,
             timesurface=0, hotpixfilter=1, hotpix_num_stds=10, logtrafo=0, gammatrafo=0, gamma=0.5, normalize_events=1,
             rand_aug=1, color_jitter=0.2)
    a.update(kw)
    return types.SimpleNamespace(**a)


def _streams(n_samples, W, H, vary, n_ev=32000, seed=5):
    from mem_amd.datasets import SyntheticEventSource
    src = SyntheticEventSource(n_ev, W, H, seed=seed, vary_extent=vary)
    evs = [src(i) for i in range(n_samples)]
    evs[1] = evs[1][:9000]                                    # a short sample (no slice draw)
    evs[2][:300, 0] = 5; evs[2][:300, 1] = 9; evs[2][:300, 3] = 1   # a hot pixel
    return evs


@pytest.mark.parametrize("data_path,is_train,W,H,vary,kw", [
    ("x/ncaltech101/", True, 240, 180, True, {}),
    ("x/ncaltech101/", False, 240, 180, True, {}),
    ("x/nimagenet_npy/", True, 640, 480, False, {"color_jitter": 0.4}),
    ("x/nimagenet_npy/", False, 640, 480, False, {"rand_aug": 0}),
    ("x/DSEC/", True, 640, 440, False, {"timesurface": 1, "logtrafo": 1}),
])
def test_full_chain_vs_oracle(data_path, is_train, W, H, vary, kw):
    from mem_amd.augment import BatchAugPipeline, ChainConfig, draw_sample
    a = _args(data_path=data_path, **kw)
    cfg = ChainConfig(a, is_train)
    cfg.canvas_max = (480, 640)
    ocfg = AC.Cfg(data_path, is_train=is_train, **{k: v for k, v in kw.items()})
    evs = _streams(6, W, H, vary)
    random.seed(11); np.random.seed(12); torch.manual_seed(13)
    draws = [draw_sample(cfg, len(e)) for e in evs]
    offs = np.concatenate([[0], np.cumsum([len(e) for e in evs])])
    ev = torch.from_numpy(np.concatenate(evs, 0)).cuda()
    out, st = BatchAugPipeline(cfg, 3)(ev, offs, draws, return_stages=True)
    assert int(st["status"].sum()) == 0
    out = out.cpu()
    for b, (e, d) in enumerate(zip(evs, draws)):
        want = AC.apply(ocfg, e, d)
        h, w = want["raster"].shape[:2]
        if "dims" in st:
            assert tuple(st["dims"][b].tolist()) == (h, w), (b, st["dims"][b].tolist(), (h, w))
            got = st["raster"][b][: 3 * h * w].view(3, h, w).permute(1, 2, 0).cpu().numpy()
        else:
            got = st["raster"][b].permute(1, 2, 0).cpu().numpy()
        assert np.array_equal(got, want["raster"]), ("raster", b)
        assert (st["resampled"][b].cpu() - want["resampled"]).abs().max().item() <= 2e-6, ("resampled", b)
        # event transforms: the hot-pixel threshold / max are data reductions -- same tolerance as test_events_gpu.
        # LogTransform is log(x + 1) in float32: for the tiny values of a sparse resized frame the rounding of x + 1
        # (2^-24) dominates, and after NormalizeEvent it is divided by the maximum of the log image
        tol = 3e-6
        if kw.get("logtrafo"):
            from oracle import transforms_t as OT
            lg = OT.log_transform(OT.remove_hot_pixels(want["resampled"].clone(), 10.0))
            tol += 1.3e-7 / float(lg[0::2].max())
        assert (st["normed"][b].cpu() - want["normed"]).abs().max().item() <= tol, ("normed", b)
        if cfg.rand_aug:
            _u8_close(st["randaug_u8"][b].cpu().numpy(), want["randaug_u8"].numpy(), ("randaug", b, d.ra), one_level=False)
            assert float(((out[b] - want["out"]).abs() > 1e-6).float().mean()) <= 2 * U8_FRAC
        else:
            assert (out[b] - want["out"]).abs().max().item() <= tol, ("out", b)


def test_ncaltech_conf_runs_unmodified(tmp_path):
    """configs/ncaltech.conf of the reference (rand_aug = 1, pt_color_jitter = 0.2, data-dependent canvases) through
    run_mem_pretraining unmodified: the data_path of the config does not exist here, so (explicit --synthetic_if_missing 1)
    seeded synthetic streams of the N-Caltech101 sensor geometry stand in (loud warning); two short epochs, finite losses, checkpoint written."""
    import json
    from mem_amd.run_mem_pretraining import get_args, main
    conf = tmp_path / "ncaltech.conf"
    # the key = value set of the reference's configs/ncaltech.conf (data, restated: dataset / preprocessing / vae / model /
    # pretraining / classification / slurm sections), so that keys this entrypoint does not know are exercised too
    keys = dict(expweek="2023-01", expname="ncaltech", slurm_job_name="ncaltech", slurm_nodes=1, slurm_cpus_per_task=8,
                gpu_num=2, gpu_vram="48G", slurm_mem="48G", slurm_time="3-00:00:00", slurm_mail_type="None",
                slurm_exclude="node1", data_path="../../../../datasets/ncaltech101/", data_set="npy", input_W=224,
                input_H=224, vae_checkpoint="", pt_checkpoint="", class_checkpoint="", vae_skip=0, pt_skip=0,
                timesurface=0, hotpixfilter=1, hotpix_num_stds=10, normalize_events=1, logtrafo=0, gammatrafo=0,
                gamma=0.5, pt_color_jitter=0.2, rand_aug=1, max_random_shift_evs=8, vae_epochs=300, vae_batch_size=192,
                vae_lr="2e-4", vae_lr_decay=0.99, vae_grad_clip="1e-2", vae_kl_loss_weight="1e-10", vae_save_ckpt_freq=25,
                vae_hidden_dim=384, vae_num_resnet_blocks=3, vae_loss="mse", vae_straight_through=0, num_tokens=8192,
                emb_dim=32, num_layers=4, transformer_depth=12, transformer_heads=12, transformer_mlp_ratio=4,
                transformer_emb=768, num_mask_patches=98, pretrained=0, masking="block", mae=0, pt_epochs=3000,
                pt_batch_size=512, pt_lr="5e-4", pt_lr_decay=0.97, pt_warmup_steps=1000, pt_grad_clip=30.0,
                pt_dropout=0.1, pt_save_ckpt_freq=25, class_epochs=300, class_update_freq=2, class_batch_size=1024,
                class_lr="4e-3", class_lr_decay=0.98, class_warmup_epochs=20, class_dropout=0.1,
                class_weight_decay="5e-2", class_save_ckpt_freq=25)
    conf.write_text("".join(f"{k} = {v}\n" for k, v in keys.items()))
    out = tmp_path / "run"
    out.mkdir()
    # command line on top of the config: a short run on a small model (explicit flags win over the config file)
    args = get_args(["--config", str(conf), "--batch_size", "8", "--epochs", "2", "--warmup_epochs", "0",
                     "--synthetic_samples", "16", "--output_dir", str(out), "--transformer_depth", "2",
                     "--transformer_emb", "128", "--transformer_heads", "2", "--num_workers", "2", "--num_tokens", "512",
                     "--warmup_steps", "-1", "--synthetic_if_missing", "1"])
    assert args.rand_aug == 1 and abs(args.color_jitter - 0.2) < 1e-12 and "ncaltech101" in args.data_path
    assert args.clip_grad == 30.0 and args.num_mask_patches == 98 and args.masking == "block"
    main(args)
    log = [json.loads(l) for l in open(out / "log.txt")]
    assert len(log) == 2 and all(np.isfinite(e["train_loss"]) for e in log)


def test_bad_samples_are_reported_at_the_deferred_check():
    """The reference raises in the transform chain (ValueError: max() of an empty array; IndexError from np.add.at) for a
    sample that is empty after the event filter or has events outside its canvas.  The batched GPU chain turns such a
    sample into a zero image and a status word; the training loop accumulates the words on the device and raises at its
    next meter flush (engine_for_pretraining.check_bad_samples) instead of training on it silently."""
    from mem_amd import engine_for_pretraining as E
    from mem_amd.augment import BatchAugPipeline, ChainConfig, draw_sample
    a = _args(data_path="x/nimagenet_npy/", rand_aug=0, color_jitter=0.0)
    cfg = ChainConfig(a, False)                                # evaluation chain: fixed 224 x 224 canvas after the rescale
    evs = _streams(4, 640, 480, False)
    random.seed(1); np.random.seed(2); torch.manual_seed(3)
    draws = [draw_sample(cfg, len(e)) for e in evs]
    offs = np.concatenate([[0], np.cumsum([len(e) for e in evs])])
    pipe = BatchAugPipeline(cfg, 3)
    E.check_bad_samples()                                      # clean start
    ev = torch.from_numpy(np.concatenate(evs, 0)).cuda()
    _, st = pipe(ev, offs, draws, return_stages=True)
    E._note_status(ev.device, st["status"])
    E.check_bad_samples()                                      # nothing flagged
    bad = [e.copy() for e in evs]
    bad[2][15000:15007, 0] = 1e6  # (inside every slice window of SliceRandomMaxEvs)
    #                                        # seven events whose FLAT index x + W * y leaves the canvas
    ev = torch.from_numpy(np.concatenate(bad, 0)).cuda()
    _, st = pipe(ev, offs, draws, return_stages=True)
    assert st["status"].tolist() == [0, 0, 1 << 28, 0] and st["bad_index_events"].tolist() == [0, 0, 7, 0]
    E._note_status(ev.device, st["status"])
    with pytest.raises(ValueError, match="1 sample"):
        E.check_bad_samples()
    E.check_bad_samples()                                      # the counter was reset
