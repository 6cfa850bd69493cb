"""-m "not gpu": the augmentation chain's oracle against the fixtures written from the REFERENCE's EventRandAugment
(tests/golden/randaug.npz, oracle/gen_golden_aug.py), and the product's host-side draw logic (mem_amd/augment.py:
draw_sample -- CPU only) against the oracle's reference-order draws under identical RNG seeds.
Reference: mem/datasets.py:611-660 (order of the chain), mem/transforms.py:441-463 (RandAugment draws)."""
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


def test_oracle_randaug_vs_reference_golden():
    g = np.load(os.path.join(GOLDEN, "randaug.npz"))
    for s in g["seeds"]:
        torch.manual_seed(int(s))
        draws = A.rand_augment_draw(224, 224, num_ops=2, magnitude=20, num_bins=31)
        assert [A.OPS.index(n) for n, _ in draws] == list(g[f"s{s}__ops"])
        assert [m for _, m in draws] == list(g[f"s{s}__mags"])
        out = A.rand_augment(event_like_u8(int(s)), draws)
        assert np.array_equal(out.numpy(), g[f"s{s}__out"]), s
    from oracle import transforms_t as T
    x = torch.from_numpy(g["tounit8__in"])
    assert np.array_equal(T.to_uint8(x).numpy(), g["tounit8__out"])
    assert np.array_equal(T.to_float32(T.to_uint8(x)).numpy(), g["tofloat32__out"])


def _args(**kw):
    a = dict(data_path="x/ncaltech101/", input_H=224, input_W=224, slice_max_evs=30000, max_random_shift_evs=8,
             timesurface=0, hotpixfilter=1, hotpix_num_stds=10, logtrafo=0, gammatrafo=0, gamma=0.5, normalize_events=1,
             rand_aug=1, color_jitter=0.2)
    a.update(kw)
    return types.SimpleNamespace(**a)


@pytest.mark.parametrize("data_path,is_train,rand_aug,cj", [("x/ncaltech101/", True, 1, 0.2), ("x/ncaltech101/", False, 1, 0.2),
                                                             ("x/nimagenet_npy/", True, 1, 0.4), ("x/nimagenet_npy/", False, 0, 0.0),
                                                             ("x/DSEC/", True, 0, 0.2), ("x/N-Cars/", True, 1, 0.0)])
def test_host_draws_follow_the_reference_order(data_path, is_train, rand_aug, cj):
    """Same seeds -> the product's draw_sample consumes random / np.random / torch exactly like the oracle's
    reference-order chain, sample after sample (including the interleaved mask generator on `random`)."""
    from mem_amd.augment import ChainConfig, draw_sample
    from mem_amd.masking_generator import MaskingGenerator
    from oracle import masking_py as MP
    import contextlib, io
    a = _args(data_path=data_path, rand_aug=rand_aug, color_jitter=cj)
    cfg = ChainConfig(a, is_train)
    ocfg = AC.Cfg(data_path, is_train=is_train, rand_aug=rand_aug, color_jitter=cj)
    assert (cfg.canvas, cfg.resize, cfg.crop, cfg.scale, cfg.rand_aug) == (ocfg.canvas, ocfg.resize, ocfg.crop, ocfg.scale, ocfg.rand_aug)
    lens = [31000, 12000, 45000, 30000, 30001]

    def seed():
        random.seed(7); np.random.seed(8); torch.manual_seed(9)
    seed()
    with contextlib.redirect_stdout(io.StringIO()):
        mg = MaskingGenerator((14, 14), 98, min_num_patches=16)
    got = []
    for n in lens:
        d = draw_sample(cfg, n)
        got.append((d, mg()))
    seed()
    with contextlib.redirect_stdout(io.StringIO()):
        om = MP.BlockMaskOracle((14, 14), 98, min_num_patches=16)
    for n, (d, mask) in zip(lens, got):
        w = AC.draws(ocfg, n)
        wm = om()
        assert (d.beg, d.end, d.time_flip, d.flip_x, d.shift, d.crop) == (w["beg"], w["end"], w["time_flip"], w["flip_x"], w["shift"], w["crop"])
        assert d.ra == w["ra"]
        assert (d.jitter[0], d.jitter[1], d.jitter[2]) == tuple(w["jitter"])
        assert np.array_equal(mask, wm)


def test_affine_identity_and_equalize_properties():
    """Sanity of the third-party restatement itself: zero-magnitude geometric ops are the identity; equalize of a
    constant channel returns it; posterize(8) and solarize(256) are identities; autocontrast stretches to [0, 255]."""
    img = event_like_u8(3)
    for name in ("ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate"):
        assert torch.equal(A.apply_op(img, name, 0.0), img), name
    assert torch.equal(A.apply_op(img, "Posterize", 8), img) and torch.equal(A.apply_op(img, "Solarize", 256.0), img)
    c = torch.full((3, 8, 8), 9, dtype=torch.uint8)
    assert torch.equal(A.equalize(c), c) and torch.equal(A.autocontrast(c), c)
    ac = A.autocontrast(img)
    assert int(ac[0].min()) == 0 and int(ac[0].max()) == 255
    t = A.apply_op(img, "TranslateX", 10.0)                      # content moves right by 10 pixels
    assert torch.equal(t[:, :, 10:], img[:, :, :-10]) and int(t[:, :, :10].max()) == 0
