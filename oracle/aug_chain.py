"""ORACLE (test infrastructure): the reference's per-sample transform chain, end to end on the CPU --
build_transformNPY (/root/reference/mem/datasets.py:611-660) + ColorJitter + mask (datasets.py:33-37,71-75).

Event-level steps come from oracle/events_np.py (pinned bit-exact against the reference classes), the tensor transforms
from oracle/transforms_t.py (pinned), the torchvision steps from oracle/aug_t.py (third party, parity unpinned; the
EventRandAugment part pinned by oracle/gen_golden_aug.py).  ``draws()`` consumes the RNGs in the reference's order;
``apply()`` replays a set of draws (also the product's SampleDraws) and returns every intermediate stage."""
import random

import numpy as np
import torch

from . import aug_t as A
from . import events_np as E
from . import transforms_t as T


class Cfg:
    """The decisions build_transformNPY takes from args (same attribute names as mem_amd.augment.ChainConfig)."""

    def __init__(self, data_path, input_H=224, input_W=224, is_train=True, slice_max_evs=30000, max_random_shift_evs=8,
                 timesurface=0, hotpixfilter=1, hotpix_num_stds=10, logtrafo=0, gammatrafo=0, gamma=0.5, normalize_events=1,
                 rand_aug=1, color_jitter=0.2):
        self.is_train, self.out_h, self.out_w = is_train, input_H, input_W
        self.slice_max, self.max_shift, self.time_surface = slice_max_evs, max_random_shift_evs, bool(timesurface)
        self.scale, self.canvas, self.resize = None, None, False
        if "imagenet" in data_path:                                           # :615-621
            H, W = input_H, input_W
            if is_train:
                s = 256 / 480
                self.scale = (s, s)
                H, W = int(480 * (256 / 480)), int(640 * (256 / 480))
            else:
                self.scale = (W / 640, H / 480)
            self.canvas = (H, W)
        elif "SS_final" in data_path or "dsec" in data_path or "DSEC" in data_path:
            self.canvas = (440, 640)
        if any(k in data_path for k in ("caltech", "Caltech", "ncars", "N-Cars", "SS_final", "dsec", "DSEC")):
            self.resize = True                                                # :638-639
        self.crop = is_train                                                  # :641-642
        self.kw = dict(timesurface=bool(timesurface), hotpixfilter=bool(hotpixfilter), num_stds=hotpix_num_stds,
                       logtrafo=bool(logtrafo), gammatrafo=bool(gammatrafo), gamma=gamma, normalize=bool(normalize_events))
        self.rand_aug = bool(is_train and rand_aug)
        self.color_jitter = color_jitter


def draws(cfg, n_events):
    """RNG consumption of ONE sample in the reference's order (datasets.py:623-658 then :36, :75 is the caller's)."""
    d = {"beg": 0, "end": n_events, "time_flip": False, "flip_x": False, "shift": None, "crop": None, "ra": None}
    if n_events > cfg.slice_max:                                              # SliceRandomMaxEvs
        start = random.choice(range(n_events - cfg.slice_max + 1))
        d["beg"], d["end"] = start, start + cfg.slice_max
    if cfg.is_train:
        d["time_flip"] = bool(np.random.random() < 0.5)
        d["flip_x"] = bool(np.random.random() < 0.5)
        xs, ys = np.random.randint(-cfg.max_shift, cfg.max_shift + 1, size=(2,))
        d["shift"] = (int(xs), int(ys))
    if cfg.crop:
        h, w = (cfg.out_h, cfg.out_w) if cfg.resize else cfg.canvas
        if A.padded_size(h, w, cfg.out_h, cfg.out_w) != (cfg.out_h, cfg.out_w):
            d["crop"] = A.random_crop_params(h, w, cfg.out_h, cfg.out_w)
    if cfg.rand_aug:
        d["ra"] = A.rand_augment_draw(cfg.out_h, cfg.out_w, num_ops=2, magnitude=20, num_bins=31)
    d["jitter"] = A.color_jitter_draw(cfg.color_jitter, cfg.color_jitter)
    return d


def apply(cfg, events, d):
    """events (N,4) f64 -> dict of stages; d = draws() dict or a mem_amd.augment.SampleDraws."""
    g = (lambda k: d[k]) if isinstance(d, dict) else (lambda k: getattr(d, k))
    st = {}
    x = np.array(events, dtype=np.float64, copy=True)
    if cfg.scale is not None:
        x[:, 0] *= cfg.scale[0]
        x[:, 1] *= cfg.scale[1]
    x = x[g("beg"):g("end")]
    H, W = cfg.canvas if cfg.canvas is not None else (None, None)
    if cfg.is_train:
        x = E.random_time_flip(x, 0.0 if g("time_flip") else 1.0)
        x = E.flip_along_x(x, 0.0 if g("flip_x") else 1.0, W=W)
        xs, ys = g("shift")
        x = E.random_shift(x, xs, ys, H, W)
    img = E.event_arr_to_img(x, H, W, cfg.time_surface)                     # (h, w, 3) u8
    st["raster"] = img
    t = A.to_tensor_u8_hwc(img)
    if cfg.resize:
        t = A.resize_bilinear_aa(t, (cfg.out_h, cfg.out_w))
    if cfg.crop:
        i, j = g("crop") or (0, 0)
        t = A.random_crop(t, cfg.out_h, cfg.out_w, i, j)
    st["resampled"] = t
    t = T.event_chain(t, **cfg.kw)
    st["normed"] = t
    if cfg.rand_aug:
        u = T.to_uint8(t)
        u = A.rand_augment(u, g("ra"))
        st["randaug_u8"] = u
        t = T.to_float32(u)
    fn_idx, bf, sf = g("jitter")
    t = A.color_jitter(t, fn_idx, bf, sf)
    st["out"] = t
    return st
