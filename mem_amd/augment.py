"""Batched on-GPU form of the reference's per-sample transform chain (build_transformNPY + ColorJitter,
/root/reference/mem/datasets.py:26-82,611-660; EventRandAugment /root/reference/mem/transforms.py:292-484).

Split of work:
  * HOST, per sample, CPU only (safe in DataLoader workers): every random DRAW of the chain, made with the same
    generators in the same order as the reference -- ``random.choice`` (slice), ``np.random.random`` x2 (time flip,
    x flip), ``np.random.randint(size=2)`` (shift), ``torch.randint`` x2 (RandomCrop, only when the size differs),
    3 x ``torch.randint`` per RandAugment op, ``torch.randperm(4)`` + ``uniform_`` x2 (ColorJitter), then the mask
    generator (``random``) -- packed into one ``SampleDraws`` record.
  * DEVICE, per batch: events (CSR in HBM) -> extent -> inferred flip / filter sizes -> extent -> per-sample canvas ->
    rasterize -> ToTensor + Resize(antialias) | RandomCrop -> fused event transforms -> ToUnit8 -> RandAugment ops ->
    ToFloat32 + ColorJitter -> f32 [B, C, H, W].  No host synchronisation anywhere in the chain (data-dependent
    canvases "W = xs.max() + 1" are resolved on the device, memhip_aug_resolve).
Geometric / photometric arithmetic = torchvision's tensor algorithms (see oracle/aug_t.py for the restatement the
tests compare against); there is no CPU fallback.
"""
import math
import random

import numpy as np
import torch

from . import transforms as T
from ._lib import C, check, declare, i32, i64, lib, ptr, stream_ptr, sz, vp

declare({
    "memhip_aug_resolve": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "memhip_rasterize_var_f64": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "memhip_resample_to_f32": (i32, [vp, vp, i64, i32, i32, i32, vp, i32, i32, i32, vp, vp]),
    "memhip_to_uint8": (i32, [vp, i64, vp, vp]),
    "memhip_rand_augment_u8": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "memhip_color_jitter": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp]),
})

INFER_FLIP_W, INFER_FILT_W, INFER_FILT_H = 1, 2, 4
OPS = ["Identity", "ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate", "Brightness", "Color", "Contrast",
       "Sharpness", "Posterize", "Solarize", "AutoContrast", "Equalize"]          # transforms.py:391-392 (small=False)
RANDAUG_DTYPE = np.dtype([("op", "<i4"), ("mag", "<f4"), ("theta", "<f4", (6,))])
JITTER_DTYPE = np.dtype([("order", "<i4"), ("bf", "<f4"), ("bf1", "<f4"), ("sf", "<f4"), ("sf1", "<f4")])
assert RANDAUG_DTYPE.itemsize == 32 and JITTER_DTYPE.itemsize == 20


def _inverse_affine_matrix(angle, translate, shear):
    """torchvision _get_inverse_affine_matrix(center=[0, 0], scale=1.0) in Python floats, like F.affine does."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / 1.0 for x in m]
    m[2] += m[0] * (-0.0 - tx) + m[1] * (-0.0 - ty)
    m[5] += m[3] * (-0.0 - tx) + m[4] * (-0.0 - ty)
    m[2] += 0.0
    m[5] += 0.0
    return m


def randaug_record(name, magnitude):
    """One (op, magnitude) of EventRandAugment -> the device record of memhip_rand_augment_u8 (transforms.py:292-330)."""
    r = np.zeros((), dtype=RANDAUG_DTYPE)
    r["op"] = OPS.index(name)
    m = None
    if name == "ShearX":
        m = _inverse_affine_matrix(0.0, [0.0, 0.0], [math.degrees(magnitude), 0.0])
    elif name == "ShearY":
        m = _inverse_affine_matrix(0.0, [0.0, 0.0], [0.0, math.degrees(magnitude)])
    elif name == "TranslateX":
        m = _inverse_affine_matrix(0.0, [1.0 * int(magnitude), 0.0], [0.0, 0.0])
    elif name == "TranslateY":
        m = _inverse_affine_matrix(0.0, [0.0, 1.0 * int(magnitude)], [0.0, 0.0])
    elif name == "Rotate":
        m = _inverse_affine_matrix(-magnitude, [0.0, 0.0], [0.0, 0.0])
    if m is not None:
        r["theta"] = np.asarray(m, dtype=np.float64).astype(np.float32)
    elif name in ("Brightness", "Color", "Contrast", "Sharpness"):
        ratio = 1.0 + magnitude                                   # F.adjust_*(img, 1.0 + magnitude)
        r["theta"][1] = np.float32(ratio)
        r["theta"][0] = np.float32(1.0 - ratio)
    elif name == "Posterize":
        r["mag"] = int(magnitude)
    elif name == "Solarize":
        r["mag"] = magnitude
    return r


def _augmentation_space(num_bins, image_size):
    """transforms.py:407-425 (all 14 ops)."""
    return {
        "Identity": (torch.tensor(0.0), False),
        "ShearX": (torch.linspace(0.0, 0.3, num_bins), True),
        "ShearY": (torch.linspace(0.0, 0.3, num_bins), True),
        "TranslateX": (torch.linspace(0.0, 150.0 / 331.0 * image_size[1], num_bins), True),
        "TranslateY": (torch.linspace(0.0, 150.0 / 331.0 * image_size[0], num_bins), True),
        "Rotate": (torch.linspace(0.0, 30.0, num_bins), True),
        "Brightness": (torch.linspace(0.0, 0.9, num_bins), True),
        "Color": (torch.linspace(0.0, 0.9, num_bins), True),
        "Contrast": (torch.linspace(0.0, 0.9, num_bins), True),
        "Sharpness": (torch.linspace(0.0, 0.9, num_bins), True),
        "Posterize": (8 - (torch.arange(num_bins) / ((num_bins - 1) / 4)).round().int(), False),
        "Solarize": (torch.linspace(255.0, 0.0, num_bins), False),
        "AutoContrast": (torch.tensor(0.0), False),
        "Equalize": (torch.tensor(0.0), False),
    }


class ChainConfig:
    """What build_transformNPY + DataAugmentationForPT decide from ``args`` (datasets.py:26-38,611-660)."""

    def __init__(self, args, is_train):
        self.is_train = bool(is_train)
        dp = args.data_path
        self.out_h, self.out_w = int(args.input_H), int(args.input_W)
        self.slice_max = int(args.slice_max_evs)
        self.max_shift = int(args.max_random_shift_evs)
        self.time_surface = bool(args.timesurface)
        self.scale = None                                    # ReshapeScaleXandY (x, y factors)
        self.canvas = None                                   # fixed (H, W) or None = inferred from the events
        self.resize = False
        if "imagenet" in dp:
            H, W = self.out_h, self.out_w
            if is_train:
                s = 256 / [480, 640][int(np.argmin([480, 640]))]                      # datasets.py:477-479
                self.scale = (s, s)
                H, W = int(480 * (256 / 480)), int(640 * (256 / 480))                 # :619-621
            else:
                self.scale = (W / 640, H / 480)
            self.canvas = (H, W)
        elif any(k in dp for k in ("SS_final", "dsec", "DSEC")):
            self.canvas = (440, 640)
        elif getattr(args, "fixed_canvas", False):           # synthetic streams generated on the model's own canvas
            self.canvas = (self.out_h, self.out_w)
        if any(k in dp for k in ("caltech", "Caltech", "ncars", "N-Cars", "SS_final", "dsec", "DSEC")):
            self.resize = True                               # :638-639
        self.crop = self.is_train                            # :641-642 RandomCrop(pad_if_needed=True)
        # canvases inferred from the data are bounded by the sensor: slot size of the var-canvas rasterizer
        self.canvas_max = (int(getattr(args, "canvas_max_H", 0)) or 480, int(getattr(args, "canvas_max_W", 0)) or 640)
        self.flags = ((0 if args.timesurface else T.EV_RM_TS) | (T.EV_HOTPIX if args.hotpixfilter else 0)
                      | (T.EV_LOG if args.logtrafo else 0) | (T.EV_GAMMA if args.gammatrafo else 0)
                      | (T.EV_NORMALIZE if args.normalize_events else 0))
        self.num_stds, self.gamma = float(args.hotpix_num_stds), float(args.gamma)
        self.rand_aug = bool(self.is_train and args.rand_aug)
        self.ra_magnitude, self.ra_bins, self.ra_ops = 20, 31, 2       # EventRandAugment(small=False, magnitude=20) :656
        self.color_jitter = float(getattr(args, "color_jitter", 0) or 0)
        self.apply_jitter = True                                       # DataAugmentationForPT always composes ColorJitter

    def pre_crop_size(self):
        """(h, w) of the image RandomCrop sees, when it is known without looking at the data."""
        if self.resize:
            return self.out_h, self.out_w
        return self.canvas


class SampleDraws:
    """Every random decision of one sample, drawn in the reference's order (see module docstring)."""
    __slots__ = ("beg", "end", "time_flip", "flip_x", "shift", "crop", "ra", "jitter")


def draw_sample(cfg: ChainConfig, n_events):
    d = SampleDraws()
    d.beg, d.end = 0, int(n_events)
    if n_events > cfg.slice_max:                                              # SliceRandomMaxEvs :494-497
        start = random.choice(range(n_events - cfg.slice_max + 1))
        d.beg, d.end = start, start + cfg.slice_max
    d.time_flip = d.flip_x = False
    d.shift = None
    if cfg.is_train:
        d.time_flip = bool(np.random.random() < 0.5)                          # RandomTimeFlip :602
        d.flip_x = bool(np.random.random() < 0.5)                             # Aug_FlipEvsAlongX :518
        xs, ys = np.random.randint(-cfg.max_shift, cfg.max_shift + 1, size=(2,))   # Aug_RandomShiftEvs :542
        d.shift = (int(xs), int(ys))
    d.crop = None
    if cfg.crop:
        hw = cfg.pre_crop_size()
        if hw is None:
            raise NotImplementedError("RandomCrop of a data-dependent canvas without Resize (no reference config does this)")
        h, w = hw
        th, tw = cfg.out_h, cfg.out_w
        ph, pw = (2 * th - h if h < th else h), (2 * tw - w if w < tw else w)  # pad_if_needed pads both sides
        if (ph, pw) != (th, tw):                                              # RandomCrop.get_params
            i = int(torch.randint(0, ph - th + 1, size=(1,)).item())
            j = int(torch.randint(0, pw - tw + 1, size=(1,)).item())
            d.crop = (i, j)
    d.ra = None
    if cfg.rand_aug:                                                          # EventRandAugment.forward :441-463
        meta = _augmentation_space(cfg.ra_bins, (cfg.out_h, cfg.out_w))
        names = list(meta.keys())
        d.ra = []
        for _ in range(cfg.ra_ops):
            name = names[int(torch.randint(len(meta), (1,)).item())]
            mags, signed = meta[name]
            r0 = torch.randint(cfg.ra_magnitude + 1, (1,)).item()
            r1 = torch.randint(2, (1,))
            m = float(mags[r0].item()) if mags.ndim > 0 else 0.0
            if signed and r1:
                m *= -1.0
            d.ra.append((name, m))
    d.jitter = None
    if cfg.apply_jitter:                                                      # torchvision ColorJitter.get_params
        fn_idx = [int(v) for v in torch.randperm(4)]
        b = s = cfg.color_jitter
        bf = float(torch.empty(1).uniform_(max(0.0, 1.0 - b), 1.0 + b)) if b else None
        sf = float(torch.empty(1).uniform_(max(0.0, 1.0 - s), 1.0 + s)) if s else None
        d.jitter = (fn_idx, bf, sf)
    return d


def jitter_record(fn_idx, bf, sf):
    r = np.zeros((), dtype=JITTER_DTYPE)
    order = 0
    if bf is not None and sf is not None:
        order = 3 if fn_idx.index(0) < fn_idx.index(2) else 4
    elif bf is not None:
        order = 1
    elif sf is not None:
        order = 2
    r["order"] = order
    bf = 1.0 if bf is None else bf
    sf = 1.0 if sf is None else sf
    r["bf"], r["bf1"], r["sf"], r["sf1"] = np.float32(bf), np.float32(1.0 - bf), np.float32(sf), np.float32(1.0 - sf)
    return r


class BatchAugPipeline:
    """events (cuda f64 [N,4]) + CSR offsets of the UNSLICED samples + per-sample draws -> f32 [B, out_chans, H, W]."""

    def __init__(self, cfg: ChainConfig, out_chans=3):
        self.cfg, self.out_chans = cfg, out_chans

    def pack(self, draws, sample_offsets):
        """Host: draws -> (window offsets i64 [B, 2] pairs flattened as CSR is not contiguous any more, aug records, ...)."""
        from .datasets import _new_aug_array
        cfg = self.cfg
        B = len(draws)
        aug = _new_aug_array(B)
        win = np.empty((B, 2), dtype=np.int64)
        for b, d in enumerate(draws):
            win[b] = (sample_offsets[b] + d.beg, sample_offsets[b] + d.end)
            if cfg.scale is not None:
                aug["scale_x"][b], aug["scale_y"][b] = cfg.scale
            if d.time_flip:
                aug["time_flip"][b] = 1
            fixed = cfg.canvas
            if d.flip_x:
                aug["flip_x"][b] = 1
                if fixed is not None:
                    aug["flip_w"][b] = fixed[1]
                else:
                    aug["infer"][b] |= INFER_FLIP_W
            if d.shift is not None:
                aug["shift_x"][b], aug["shift_y"][b] = d.shift
                aug["do_filter"][b] = 1
                if fixed is not None:
                    aug["filt_w"][b], aug["filt_h"][b] = fixed[1], fixed[0]
                else:
                    aug["infer"][b] |= INFER_FILT_W | INFER_FILT_H
        crop = None
        if cfg.crop and any(d.crop is not None for d in draws):
            crop = np.zeros((B, 2), dtype=np.int32)
            for b, d in enumerate(draws):
                if d.crop is not None:
                    crop[b] = d.crop
        ra = None
        if cfg.rand_aug:
            ra = np.zeros((cfg.ra_ops, B), dtype=RANDAUG_DTYPE)
            for b, d in enumerate(draws):
                for k, (name, m) in enumerate(d.ra):
                    ra[k, b] = randaug_record(name, m)
        jit = None
        if cfg.apply_jitter and cfg.color_jitter:
            jit = np.zeros(B, dtype=JITTER_DTYPE)
            for b, d in enumerate(draws):
                jit[b] = jitter_record(*d.jitter)
        return win, aug, crop, ra, jit

    def __call__(self, ev, sample_offsets, draws, return_stages=False):
        """ev: cuda f64 [N,4] (all samples concatenated); sample_offsets: host int sequence [B+1]."""
        cfg = self.cfg
        dev = ev.device
        B = len(draws)
        win, aug_h, crop_h, ra_h, jit_h = self.pack(draws, sample_offsets)
        up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(dev, non_blocking=True)   # noqa: E731
        # per-sample windows are [beg, end) pairs: the kernels read offsets[b], offsets[b+1] -> give every sample its
        # own 2-entry CSR by launching on an interleaved array when windows are not contiguous
        contiguous = bool(np.all(win[1:, 0] == win[:-1, 1]))
        aug = up(aug_h)
        st = stream_ptr()
        stages = {}
        from .datasets import events_extent, rasterize
        if contiguous:
            offs = torch.from_numpy(np.concatenate([win[:, 0], win[-1:, 1]])).to(dev, non_blocking=True)
            ev_w = ev
        else:
            # gather the windows into one contiguous event buffer (device-to-device row copies, 32 B per event)
            idx = torch.from_numpy(np.concatenate([np.arange(a, b, dtype=np.int64) for a, b in win])).to(dev, non_blocking=True)
            ev_w = ev.index_select(0, idx)
            lens = win[:, 1] - win[:, 0]
            offs = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(dev, non_blocking=True)
        status = torch.zeros(B, dtype=torch.int32, device=dev)
        if cfg.canvas is not None:
            H, W = cfg.canvas
            st_r = []
            img = rasterize(ev_w, offs, H, W, cfg.time_surface, aug, strict=False, status_out=st_r)
            st2 = st_r[0]
            dims, slot = None, 3 * H * W
        else:
            Hm, Wm = cfg.canvas_max
            ext = events_extent(ev_w, offs, None)
            check(lib.memhip_aug_resolve(ptr(ext), ptr(aug), B, 0, 0, 0, Hm, Wm, None, None, st), "aug_resolve")
            ext = events_extent(ev_w, offs, aug)
            dims = torch.empty((B, 2), dtype=torch.int32, device=dev)
            check(lib.memhip_aug_resolve(ptr(ext), ptr(aug), B, 1, 0, 0, Hm, Wm, ptr(dims), ptr(status), st), "aug_resolve")
            img = torch.empty((B, 3 * Hm * Wm), dtype=torch.uint8, device=dev)
            wsb = lib.memhip_rasterize_workspace(B, Hm, Wm)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            st2 = torch.empty(B, dtype=torch.int32, device=dev)
            check(lib.memhip_rasterize_var_f64(ptr(ev_w), ptr(offs), ptr(aug), ptr(dims), B, Hm, Wm, int(cfg.time_surface),
                                               ptr(img), ptr(st2), ptr(ws), wsb, st), "rasterize_var")
            H, W, slot = 0, 0, 3 * Hm * Wm
            stages["dims"] = dims
        stages["raster"] = img
        OH, OW = cfg.out_h, cfg.out_w
        x = torch.empty((B, 3, OH, OW), dtype=torch.float32, device=dev)
        if cfg.resize:
            check(lib.memhip_resample_to_f32(ptr(img), ptr(dims), slot, H, W, 0, None, B, OH, OW, ptr(x), st), "resize")
        else:
            crop = up(crop_h) if crop_h is not None else None
            if dims is None and (H, W) != (OH, OW) and not cfg.crop:
                raise ValueError(f"canvas {H}x{W} != model input {OH}x{OW} and the chain has neither Resize nor RandomCrop")
            check(lib.memhip_resample_to_f32(ptr(img), ptr(dims), slot, H, W, 1, ptr(crop), B, OH, OW, ptr(x), st), "crop")
        stages["resampled"] = x
        x = T.event_norm(x, cfg.flags, cfg.num_stds, cfg.gamma, 3)
        stages["normed"] = x
        src, src_u8 = x, False
        if cfg.rand_aug:
            a = torch.empty((B, 3, OH, OW), dtype=torch.uint8, device=dev)
            b2 = torch.empty_like(a)
            check(lib.memhip_to_uint8(ptr(x), x.numel(), ptr(a), st), "to_uint8")
            ra = up(ra_h).view(cfg.ra_ops, -1)
            for k in range(cfg.ra_ops):
                check(lib.memhip_rand_augment_u8(ptr(a), ptr(b2), ptr(ra[k]), B, OH, OW, st), "rand_augment")
                a, b2 = b2, a
            stages["randaug_u8"] = a
            src, src_u8 = a, True
        jit = up(jit_h) if jit_h is not None else None
        out = torch.empty((B, self.out_chans, OH, OW), dtype=torch.float32, device=dev)
        check(lib.memhip_color_jitter(ptr(src), int(src_u8), B, OH, OW, ptr(jit), ptr(out), self.out_chans, st), "color_jitter")
        # bit 29: empty sample after the filter / canvas beyond canvas_max (the reference: ValueError from max() of an
        # empty array); bit 28: events outside the canvas (the reference: IndexError from np.add.at).  Such a sample is an
        # all-zero image here; callers check `status` (TransformNPY at once, the training loop at its meter flush).
        stages["status"] = status | ((st2 != 0).to(torch.int32) << 28)
        stages["bad_index_events"] = st2
        return (out, stages) if return_stages else out
