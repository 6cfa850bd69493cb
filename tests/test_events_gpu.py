"""-m gpu: HIP rasterizer / fused event augs / event_norm vs the oracle and the reference goldens
(bit-exact: integer voxel bins; fp32 transforms with exact zeroing)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _ev_dev(ev):
    return torch.from_numpy(np.ascontiguousarray(ev, dtype=np.float64)).cuda()


def _off(*ns):
    return torch.tensor(np.concatenate([[0], np.cumsum(ns)]), dtype=torch.int64, device="cuda")


def test_rasterizer_reference_goldens():
    from mem_amd import datasets as D
    g = np.load(os.path.join(GOLDEN, "events_raster.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "events_raster.json")))
    for m in meta:
        ev = g[m["name"] + "__ev"]
        want = g[m["name"] + "__img"]
        got = D.EventArrToImg(m["H"], m["W"], m["timesurface"])(ev.copy())   # reference call shape
        assert got.dtype == np.uint8 and got.shape == want.shape, m["name"]
        assert np.array_equal(got, want), m["name"]
    with pytest.raises(IndexError):
        D.EventArrToImg(100, 120, False)(g["oob__ev"].copy())


def test_rasterizer_batched_ragged_vs_oracle():
    from mem_amd import datasets as D
    from oracle import events_np as E
    rng = np.random.default_rng(5)
    H, W = 224, 224
    ns = [30000, 1, 0, 12345, 30000, 7, 64, 65]          # ragged incl. an empty sample
    evs = []
    for n in ns:
        evs.append(np.stack([rng.integers(0, W, n), rng.integers(0, H, n), np.sort(rng.integers(0, 300000, n)),
                             rng.integers(0, 2, n) * 2 - 1], 1).astype(np.float64).reshape(n, 4))
    evs[0][:700, :2] = (3, 4); evs[0][:700, 3] = -1       # > 255 hits
    allv = np.concatenate(evs)
    for ts in (False, True):
        got = D.rasterize(_ev_dev(allv), _off(*ns), H, W, ts).cpu().numpy()
        for b, n in enumerate(ns):
            if n == 0:
                assert got[b].sum() == 0
                continue
            if ts and np.ptp(evs[b][:, 2]) == 0:
                continue                                  # 0/0 time surface: reference yields NaN casts
            want = E.event_arr_to_img(evs[b], H, W, ts).transpose(2, 0, 1)
            assert np.array_equal(got[b], want), (ts, b)


def test_fused_augs_vs_oracle_chain():
    """slice -> time flip -> x flip -> shift+filter -> rasterize, all in one HIP pass, against the
    oracle's five NumPy passes with the same draws."""
    from mem_amd import datasets as D
    from oracle import events_np as E
    rng = np.random.default_rng(9)
    n, H, W = 40000, 180, 240
    ev = np.stack([rng.integers(0, W, n), rng.integers(0, H, n), np.sort(rng.integers(0, 300000, n)),
                   rng.integers(0, 2, n) * 2 - 1], 1).astype(np.float64)
    for trial in range(6):
        ts = bool(trial % 2)
        seed = 100 + trial
        # --- oracle chain with explicit draws
        random.seed(seed); np.random.seed(seed)
        start = random.choice(range(n - 30000 + 1))
        x = E.slice_random_max_evs(ev, 30000, start)
        x = E.random_time_flip(x.copy(), np.random.random())
        x = E.flip_along_x(x, np.random.random(), W=None if trial < 3 else W)
        sx, sy = np.random.randint(-8, 9, size=(2,))
        x = E.random_shift(x, sx, sy, None if trial < 3 else H, None if trial < 3 else W)
        want = E.event_arr_to_img(x, None if trial < 3 else H, None if trial < 3 else W, ts)
        # --- product chain (reference class surface), same seeds
        random.seed(seed); np.random.seed(seed)
        HW = (None, None) if trial < 3 else (H, W)
        chain = D.Compose([D.SliceRandomMaxEvs(30000), D.RandomTimeFlip(), D.Aug_FlipEvsAlongX(*HW),
                           D.Aug_RandomShiftEvs(*HW, max_shift=8), D.EventArrToImg(*HW, ts)])
        got = chain(ev.copy())
        assert got.shape == want.shape and np.array_equal(got, want), trial


def test_aug_goldens_from_reference():
    from mem_amd import datasets as D
    from oracle import events_np as E
    g = np.load(os.path.join(GOLDEN, "events_augs.npz"))
    small = g["small__in"]
    # reference outputs (N',4) arrays; the fused kernel's image of them must equal the image of the golden
    for tag, HW in (("a", (None, None)), ("b", (180, 240)), ("c", (None, None))):
        xs, ys = g[f"shift_{tag}__xy"]
        want = E.event_arr_to_img(g[f"shift_{tag}__out"], 200, 260, True)
        s = D.EventStream(small.copy())
        W = HW[1] or int(small[:, 0].max()) + 1
        H = HW[0] or int(small[:, 1].max()) + 1
        s.aug["shift_x"], s.aug["shift_y"], s.aug["do_filter"], s.aug["filt_w"], s.aug["filt_h"] = xs, ys, 1, W, H
        got = D.EventArrToImg(200, 260, True)(s)
        assert np.array_equal(got, want), tag
    want = E.event_arr_to_img(g["timeflip_flip__out"], 200, 260, True)
    s = D.EventStream(small.copy()); s.aug["time_flip"] = 1
    assert np.array_equal(D.EventArrToImg(200, 260, True)(s), want)
    for tr in (0, 1):
        want = E.event_arr_to_img(g[f"rescale_{tr}__out"], 300, 400, False)
        s = D.ReshapeScaleXandY(224, 224, 480, 640, is_train=bool(tr))(g[f"rescale_{tr}__in"].copy())
        assert np.array_equal(D.EventArrToImg(300, 400, False)(s), want)


def test_event_norm_reference_goldens():
    from mem_amd import transforms as T
    g = np.load(os.path.join(GOLDEN, "transforms.npz"))
    for name in ("s32", "s224", "zeros"):
        x = torch.from_numpy(g[name + "__in"])
        assert torch.equal(T.RemoveTimesurface()(x.clone()), torch.from_numpy(g[name + "__rm_ts"]))
        r = torch.from_numpy(g[name + "__rm_ts"])
        assert torch.equal(T.RemoveHotPixels(10)(r.clone()), torch.from_numpy(g[name + "__hot10"]))
        assert torch.equal(T.RemoveHotPixels(3)(r.clone()), torch.from_numpy(g[name + "__hot3"]))
        assert torch.equal(T.NormalizeEvent()(torch.from_numpy(g[name + "__hot10"])),
                           torch.from_numpy(g[name + "__norm"]))
        assert torch.equal(T.EventChain()(x.clone()), torch.from_numpy(g[name + "__norm"]))
        # log / gamma: torch-CPU (SLEEF) vs OCML may differ in the last ulp
        torch.testing.assert_close(T.LogTransform()(r.clone()), torch.from_numpy(g[name + "__log"]), rtol=3e-7, atol=1e-7)
        torch.testing.assert_close(T.GammaTransform(0.5)(r.clone()), torch.from_numpy(g[name + "__gamma"]), rtol=3e-7, atol=0)
        n = torch.from_numpy(g[name + "__norm"])
        assert torch.equal(T.ToUnit8()(n), torch.from_numpy(g[name + "__u8"]))
        assert torch.equal(T.ToFloat32()(torch.from_numpy(g[name + "__u8"])), torch.from_numpy(g[name + "__f32"]))


def test_hot_pixel_topk_vs_reference_goldens_and_tie_properties():
    """a4, RemoveHotPixels(num_hot_pixels=k) (transforms.py:257-263): equal to the REFERENCE's outputs wherever the
    reference defines them (tie-free selection boundary; incl. k = 0, the sum / 4 clamp, a pixel hot in both
    polarities, a non-square canvas); with ties at the boundary the result must still be A valid top-k selection:
    exactly k flat entries chosen, none of the others larger than the smallest chosen one."""
    from mem_amd import transforms as T
    from oracle import transforms_t as OT
    g = np.load(os.path.join(GOLDEN, "transforms_topk.npz"))
    for name in ("t32", "t224", "t40x56"):
        x = torch.from_numpy(g[name + "__in"])
        for k in g[name + "__ks"].tolist():
            got = T.RemoveHotPixels(num_hot_pixels=k)(x.clone())
            assert torch.equal(got, torch.from_numpy(g[name + f"__top{k}"])), (name, k)
    # batched u8 input + the rest of the chain (normalise, 2-bin view) vs the oracle chain with the top-k filter
    rng = np.random.default_rng(5)
    B, H, W = 3, 224, 224
    img = torch.from_numpy(rng.integers(0, 3, (B, 3, H, W)).astype(np.uint8))
    for b in range(B):
        pos = rng.choice(H * W, 30, replace=False)
        img[b, 0].view(-1)[torch.from_numpy(pos[:15])] = torch.arange(100, 115, dtype=torch.uint8)
        img[b, 2].view(-1)[torch.from_numpy(pos[15:])] = torch.arange(120, 135, dtype=torch.uint8)
    out = T.event_norm(img.cuda(), T.EV_RM_TS | T.EV_NORMALIZE, out_chans=2, num_hot_pixels=12).cpu()
    for b in range(B):
        xf = img[b].float() / 255
        assert OT.topk_is_tie_free(xf, 12)
        want = OT.normalize_event(OT.remove_hot_pixels_topk(OT.remove_timesurface(xf), 12))
        assert torch.equal(out[b], want[0::2]), b
    # ties at the boundary (many equal counts): a valid selection, deterministic
    x = torch.zeros(3, 64, 64)
    gen = torch.Generator().manual_seed(9)
    x[0] = torch.randint(0, 6, (64, 64), generator=gen).float() / 255
    x[2] = torch.randint(0, 6, (64, 64), generator=gen).float() / 255
    for k in (1, 9, 200):
        kk = OT.topk_clamped(x, k)
        y = T.RemoveHotPixels(num_hot_pixels=k)(x.clone())
        assert torch.equal(y, T.RemoveHotPixels(num_hot_pixels=k)(x.clone()))
        flat, zeroed = x[0::2].flatten(), (y[0::2].flatten() == 0) & (x[0::2].flatten() != 0)
        vals = torch.sort(flat).values
        kth = vals[len(vals) - kk]
        # every entry strictly above the k-th value is zeroed; nothing below it is zeroed except as the other polarity
        # of a chosen pixel; the number of chosen pixels is between k / 2 and k
        assert bool((y[0::2].flatten()[flat > kth] == 0).all())
        changed_px = ((y[0] != x[0]) | (y[2] != x[2])).sum().item()
        assert changed_px <= kk
        pol_max = torch.maximum(x[0], x[2])[(y[0] != x[0]) | (y[2] != x[2])]
        assert bool((pol_max >= kth).all())


def test_event_norm_u8_batched_2chan_vs_oracle():
    from mem_amd import datasets as D, transforms as T
    from oracle import events_np as E, transforms_t as OT
    rng = np.random.default_rng(3)
    B, n, H, W = 5, 30000, 224, 224
    evs = [np.stack([rng.integers(0, W, n), rng.integers(0, H, n), np.sort(rng.integers(0, 300000, n)),
                     rng.integers(0, 2, n) * 2 - 1], 1).astype(np.float64) for _ in range(B)]
    evs[1][:500, :2] = (10, 20)                         # hot pixel in sample 1
    img = D.rasterize(_ev_dev(np.concatenate(evs)), _off(*([n] * B)), H, W, False)
    out2 = T.event_norm(img, T.EV_RM_TS | T.EV_HOTPIX | T.EV_NORMALIZE, 10.0, 0.5, 2).cpu()
    out3 = T.event_norm(img, T.EV_RM_TS | T.EV_HOTPIX | T.EV_NORMALIZE, 10.0, 0.5, 3).cpu()
    for b in range(B):
        x = torch.from_numpy(E.to_tensor_chw(E.event_arr_to_img(evs[b], H, W, False)))
        want = OT.event_chain(x)
        assert torch.equal(out3[b], want), b
        assert torch.equal(out2[b], want[0::2]), b      # the 2-bin voxel view x[:, 0::2]
    assert out3[1, 0, 20, 10] == 0                       # hot pixel removed


def test_pipeline_full_size_properties():
    """BASELINE-size batch (256 x 30k events @224^2): size-independent properties -- total count
    conservation mod 256 is not meaningful, so check per-sample sums against a torch bincount."""
    from mem_amd import datasets as D
    B, n, H, W = 256, 30000, 224, 224
    g = torch.Generator(device="cuda").manual_seed(1)
    xy = torch.randint(0, 224, (B * n, 2), generator=g, device="cuda").double()
    t = torch.rand((B * n, 1), generator=g, device="cuda", dtype=torch.float64) * 3e5
    p = (torch.randint(0, 2, (B * n, 1), generator=g, device="cuda") * 2 - 1).double()
    ev = torch.cat([xy, t, p], 1).contiguous()
    off = torch.arange(0, B + 1, device="cuda", dtype=torch.int64) * n
    img = D.rasterize(ev, off, H, W, False)
    flat = (xy[:, 0] + W * xy[:, 1]).long() + (torch.arange(B, device="cuda").repeat_interleave(n) * H * W)
    pos = torch.bincount(flat[p[:, 0] == 1], minlength=B * H * W).view(B, H, W)
    neg = torch.bincount(flat[p[:, 0] == -1], minlength=B * H * W).view(B, H, W)
    assert torch.equal(img[:, 0].long(), pos % 256) and torch.equal(img[:, 2].long(), neg % 256)
    assert int(img[:, 1].sum()) == 0
    # idempotence / linearity-like property: rasterizing the concatenation of two halves ==
    # (sum of the halves' images) mod 256
    h1 = D.rasterize(ev[: B * n // 2], off[: B // 2 + 1], H, W, False)
    assert torch.equal(h1, img[: B // 2])


def test_binned_rasterizer_vs_oracle_ragged_and_edges():
    """The two-pass long-stream kernels: ragged batch, chunk-boundary sizes, fractional / negative-wrapping
    coordinates, p == 0, events outside the canvas (status), fused augmentations -- bit-equal to the oracle
    and to the single-pass kernels."""
    from mem_amd import datasets as D
    from oracle import events_np as E
    rng = np.random.default_rng(11)
    for (H, W) in ((480, 640), (224, 224), (37, 1000)):
        ns = [4096, 4097, 0, 1, 50000, 4095, 8192, 12289]
        evs = []
        for n in ns:
            ev = np.stack([rng.uniform(0, W, n), rng.uniform(0, H, n), np.sort(rng.integers(0, 300000, n)),
                           rng.integers(-1, 2, n)], 1).astype(np.float64).reshape(n, 4)      # p in {-1, 0, 1}
            evs.append(ev)
        evs[4][:900, :2] = (3.7, 4.2); evs[4][:900, 3] = 1          # > 255 hits on one pixel
        evs[4][900:1300, :2] = (3.2, 4.9); evs[4][900:1300, 3] = -1
        evs[7][:50, 1] = -1.0 - rng.integers(0, H - 1, 50)          # negative flat index: NumPy wrap
        allv = np.concatenate(evs)
        got = D.rasterize(_ev_dev(allv), _off(*ns), H, W, False, binned=True).cpu().numpy()
        ref = D.rasterize(_ev_dev(allv), _off(*ns), H, W, False, binned=False).cpu().numpy()
        assert np.array_equal(got, ref), (H, W)
        for b, n in enumerate(ns):
            want = E.event_arr_to_img(evs[b], H, W, False).transpose(2, 0, 1) if n else np.zeros((3, H, W), np.uint8)
            assert np.array_equal(got[b], want), (H, W, b)
        # the band count is chosen from the batch size (csrc/raster.hip::choose_bands): every count gives the same image
        from mem_amd import _lib
        try:
            for bands in (1, 2, 10, 13, 16, 40, 64):
                _lib.set_option("raster_bands", bands)               # (counts below the canvas' minimum are raised to it)
                again = D.rasterize(_ev_dev(allv), _off(*ns), H, W, False, binned=True).cpu().numpy()
                assert np.array_equal(again, ref), (H, W, bands)
        finally:
            _lib.set_option("raster_bands", 0)
    # out-of-canvas events raise like the reference, on both paths
    bad = np.array([[5.0, 3.0, 0.0, 1.0], [10.0, 480.0, 1.0, 1.0]])
    for binned in (True, False):
        with pytest.raises(IndexError):
            D.rasterize(_ev_dev(bad), _off(2), 480, 640, False, binned=binned)
    # fused augmentations go through the same event decode
    n, H, W = 20000, 480, 640
    ev = np.stack([rng.integers(0, W, n), rng.integers(0, H, n), np.sort(rng.integers(0, 300000, n)),
                   rng.integers(0, 2, n) * 2 - 1], 1).astype(np.float64)
    s = D.EventStream(ev.copy())
    s.aug["time_flip"], s.aug["flip_x"], s.aug["flip_w"] = 1, 1, W
    s.aug["shift_x"], s.aug["shift_y"], s.aug["do_filter"], s.aug["filt_w"], s.aug["filt_h"] = -7, 5, 1, W, H
    off, aug = s._dev()
    a = D.rasterize(s.ev, off, H, W, False, aug, binned=True)
    b = D.rasterize(s.ev, off, H, W, False, aug, binned=False)
    assert torch.equal(a, b) and int(a.sum()) > 0


def test_binned_rasterizer_16bit_carry():
    """Counters are [neg:16 | pos:16] per pixel: 65536+ positive events on one pixel must not leak into the
    negative count (the carry is taken back out)."""
    from mem_amd import datasets as D
    n_pos, n_neg = 3 * 65536 + 77, 65536 + 300
    ev = np.zeros((n_pos + n_neg + 10, 4))
    ev[:, 0], ev[:, 1] = 17, 9
    ev[:n_pos, 3] = 1
    ev[n_pos:n_pos + n_neg, 3] = -1
    ev[n_pos + n_neg:, 0] = 18; ev[n_pos + n_neg:, 3] = -1          # the neighbour pixel stays clean
    rng = np.random.default_rng(0)
    ev = ev[rng.permutation(len(ev))]
    img = D.rasterize(_ev_dev(ev), _off(len(ev)), 480, 640, False, binned=True)[0].cpu().numpy()
    assert img[0, 9, 17] == n_pos % 256 and img[2, 9, 17] == n_neg % 256
    assert img[2, 9, 18] == 10 and img[0, 9, 18] == 0
    assert int(img.astype(np.int64).sum()) == n_pos % 256 + n_neg % 256 + 10


def test_binned_rasterizer_config4_properties():
    """BASELINE configs[3] size: ~1 M events per sample on a 480 x 640 canvas (uniform, and the hot-pixel
    variant: 1 % of the events on 16 pixels) against torch.bincount, mod 256."""
    from mem_amd import datasets as D
    B, n, H, W = 8, 1_000_000, 480, 640
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.randint(0, W, (B * n,), generator=g, device="cuda")
    y = torch.randint(0, H, (B * n,), generator=g, device="cuda")
    hot = torch.rand((B * n,), generator=g, device="cuda") < 0.01
    hp = torch.randint(0, 16, (B * n,), generator=g, device="cuda")
    x = torch.where(hot, 100 + 7 * hp, x); y = torch.where(hot, 50 + 3 * hp, y)
    t = torch.rand((B * n,), generator=g, device="cuda", dtype=torch.float64) * 3e5
    p = torch.randint(0, 2, (B * n,), generator=g, device="cuda") * 2 - 1
    ev = torch.stack([x.double(), y.double(), t, p.double()], 1).contiguous()
    off = torch.arange(0, B + 1, device="cuda", dtype=torch.int64) * n
    img = D.rasterize(ev, off, H, W, False)                         # size-based choice -> binned path
    flat = x + W * y + torch.arange(B, device="cuda").repeat_interleave(n) * H * W
    pos = torch.bincount(flat[p == 1], minlength=B * H * W).view(B, H, W)
    neg = torch.bincount(flat[p == -1], minlength=B * H * W).view(B, H, W)
    assert int(pos.max()) > 255                                      # the hot pixels wrap
    assert torch.equal(img[:, 0].long(), pos % 256) and torch.equal(img[:, 2].long(), neg % 256)
    assert int(img[:, 1].sum()) == 0
    assert torch.equal(img, D.rasterize(ev, off, H, W, False, binned=False))

def test_binned_rasterizer_key_workspace_too_small_is_flagged_per_sample():
    """include/memhip.h: `n_events` sizes the key workspace; a sample whose rows would not fit gets status |= 1 << 30 and a
    zero image, the samples in front of it are rasterized as usual (status = its count of out-of-canvas events, written by
    pass 2 from the per-workgroup slots: the caller zeroes nothing)."""
    from mem_amd import datasets as D
    from mem_amd._lib import lib, ptr, stream_ptr
    H, W = 480, 640
    rng = np.random.default_rng(5)
    ns = [5000, 7000, 300]
    evs = [np.stack([rng.integers(0, W, n), rng.integers(0, H, n), np.sort(rng.integers(0, 300000, n)), rng.integers(0, 2, n) * 2 - 1],
                    1).astype(np.float64) for n in ns]
    evs[0][:3, 1] = H + 5                                            # three events below the canvas: IndexError in the reference
    ev, off = _ev_dev(np.concatenate(evs)), _off(*ns)
    n_cap = 9000                                                     # sample 0 fits (5000), sample 1 ends at 12000, sample 2 at 12300
    wsb = lib.memhip_rasterize_binned_workspace(3, H, W, n_cap)
    ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")
    out = torch.full((3, 3, H, W), 7, dtype=torch.uint8, device="cuda")
    status = torch.full((3,), -1, dtype=torch.int32, device="cuda")
    rc = lib.memhip_rasterize_binned_f64(ptr(ev), ptr(off), None, 3, H, W, n_cap, ptr(out), ptr(status), ptr(ws), wsb, stream_ptr())
    assert rc == 0
    st = status.cpu().tolist()
    assert st == [3, 1 << 30, 1 << 30], st
    assert int(out[1:].sum()) == 0
    ok = evs[0][3:]
    want = D.rasterize(_ev_dev(ok), _off(len(ok)), H, W, False, binned=False)[0]
    assert torch.equal(out[0], want)
