"""Event data path -- mirror of /root/reference/mem/datasets.py (event classes :464-660,
DataAugmentationForPT :26-82, build_pretraining_dataset :146-174).

MI355X design: the reference runs five NumPy passes per sample in DataLoader workers
(slice -> time flip -> x flip -> shift+filter -> np.add.at).  Here the event-level classes
only *record* their parameters on a lazy ``EventStream`` (events live in HBM as the
reference's (N,4) float64 rows); ``EventArrToImg`` then runs ONE fused HIP pass
(csrc/raster.hip) that applies the whole chain while it reads each event once.  The random
draws are made on the host with the same generators, in the same order, as the reference
(``random.choice``, ``np.random.random``, ``np.random.randint``), so results are equal draw
for draw.  ``EventBatchPipeline`` is the batched form used by training / bench.
"""
import ctypes as C
import random

import numpy as np
import torch

from . import transforms as T
from ._lib import check, declare, i32, lib, ptr, require_gpu, stream_ptr, sz, vp
from .masking_generator import MaskingGenerator, MaskingGeneratorRandomLocation

declare({
    "memhip_rasterize_workspace": (sz, [i32, i32, i32]),
    "memhip_rasterize_f64": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "memhip_rasterize_aug_f64": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
    "memhip_rasterize_binned_workspace": (sz, [i32, i32, i32, C.c_int64]),
    "memhip_rasterize_binned_f64": (i32, [vp, vp, vp, i32, i32, i32, C.c_int64, vp, vp, vp, sz, vp]),
    "memhip_events_extent": (i32, [vp, vp, vp, i32, vp, vp]),
})

# rasterize() uses the two-pass binned kernels (csrc/raster.hip) whenever there is no time surface and the canvas is
# within their band limit; the time surface keeps the global-atomic form; binned=False forces the single-pass kernels
_BINNED_MAX_PIXELS = 64 * 32764



class EventAug(C.Structure):
    """== memhip_event_aug_t (include/memhip.h)."""
    _fields_ = [("scale_x", C.c_double), ("scale_y", C.c_double), ("time_flip", C.c_int32),
                ("flip_x", C.c_int32), ("flip_w", C.c_int64), ("shift_x", C.c_int32),
                ("shift_y", C.c_int32), ("do_filter", C.c_int32), ("filt_w", C.c_int32),
                ("filt_h", C.c_int32), ("pad_", C.c_int32)]

    def __init__(self):
        super().__init__()
        self.scale_x = self.scale_y = 1.0


AUG_DTYPE = np.dtype([("scale_x", "<f8"), ("scale_y", "<f8"), ("time_flip", "<i4"), ("flip_x", "<i4"),
                      ("flip_w", "<i8"), ("shift_x", "<i4"), ("shift_y", "<i4"), ("do_filter", "<i4"),
                      ("filt_w", "<i4"), ("filt_h", "<i4"), ("pad_", "<i4")])
assert AUG_DTYPE.itemsize == C.sizeof(EventAug) == 56


def _new_aug_array(n):
    a = np.zeros(n, dtype=AUG_DTYPE)
    a["scale_x"] = 1.0
    a["scale_y"] = 1.0
    return a


def rasterize(ev, offsets, H, W, time_surface=False, aug=None, strict=True, binned=None):
    """ev f64 [n,4] (cuda), offsets i64 [B+1] (cuda), aug = uint8 cuda view of B aug records or
    None -> u8 [B,3,H,W] (cuda).  strict: raise IndexError like the reference when an event lands
    outside the canvas (costs one host sync).  binned: None = choose by size, True/False = force the
    two-pass long-stream kernels / the single-pass kernels."""
    require_gpu()
    B = offsets.numel() - 1
    out = torch.empty((B, 3, H, W), dtype=torch.uint8, device=ev.device)
    status = torch.empty((B,), dtype=torch.int32, device=ev.device)
    n_rows = int(ev.shape[0])
    if binned is None:
        # measured on MI355X: the two-pass kernels beat the single-pass LDS kernel at every size tried (256 x 30 000
        # events on 224 x 224: 69 us vs 145 us -- the single pass re-reads the events once per band of the canvas)
        binned = not time_surface and B > 0 and H * W <= _BINNED_MAX_PIXELS
    if binned:
        # n_rows bounds offsets[B] - offsets[0] without a host sync
        wsb = lib.memhip_rasterize_binned_workspace(B, H, W, n_rows)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=ev.device)
        check(lib.memhip_rasterize_binned_f64(ptr(ev), ptr(offsets), ptr(aug), B, H, W, n_rows, ptr(out), ptr(status),
                                              ptr(ws), wsb, stream_ptr()), "rasterize_binned")
    else:
        wsb = lib.memhip_rasterize_workspace(B, H, W)
        ws = torch.empty((wsb,), dtype=torch.uint8, device=ev.device)
        check(lib.memhip_rasterize_aug_f64(ptr(ev), ptr(offsets), ptr(aug), B, H, W, int(bool(time_surface)),
                                           ptr(out), ptr(status), ptr(ws), wsb, stream_ptr()), "rasterize")
    if strict and int(status.sum().item()) != 0:
        raise IndexError("event outside the H x W canvas (reference: np.add.at IndexError)")
    return out


def events_extent(ev, offsets, aug=None):
    """-> f64 [B,4] (max x, max y, min x, min y) on the device."""
    B = offsets.numel() - 1
    ext = torch.empty((B, 4), dtype=torch.float64, device=ev.device)
    check(lib.memhip_events_extent(ptr(ev), ptr(offsets), ptr(aug), B, ptr(ext), stream_ptr()), "events_extent")
    return ext


class EventStream:
    """(N,4) float64 events in HBM + the augmentation record that the fused rasterizer applies."""

    def __init__(self, x):
        require_gpu()
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
        self.ev = x.to("cuda", dtype=torch.float64).contiguous()
        self.beg, self.end = 0, self.ev.shape[0]
        self.aug = _new_aug_array(1)

    def __len__(self):
        return self.end - self.beg

    def _dev(self):
        off = torch.tensor([self.beg, self.end], dtype=torch.int64, device="cuda")
        aug = torch.from_numpy(self.aug.view(np.uint8).copy()).to("cuda")
        return off, aug

    def max_xy(self):
        off, aug = self._dev()
        e = events_extent(self.ev, off, aug)[0].tolist()
        return e[0], e[1]


def _as_stream(x):
    return x if isinstance(x, EventStream) else EventStream(x)


class ReshapeScaleXandY:
    """datasets.py:464-485."""

    def __init__(self, newH=224, newW=224, oldH=480, oldW=640, is_train=False):
        assert 100 <= newH <= 640 and 100 <= newW <= 640 and 100 <= oldH <= 640 and 100 <= oldW <= 640
        if is_train:
            scale = 256 / [oldH, oldW][int(np.argmin([oldH, oldW]))]
            self.scale_x = self.scale_y = scale
        else:
            self.scale_x, self.scale_y = newW / oldW, newH / oldH

    def __call__(self, x):
        s = _as_stream(x)
        s.aug["scale_x"] *= self.scale_x
        s.aug["scale_y"] *= self.scale_y
        return s


class SliceRandomMaxEvs:
    """datasets.py:488-498: random.choice over the admissible window starts."""

    def __init__(self, keep_max_num_evs=30000):
        self.keep_max_N_evs = keep_max_num_evs
        assert 5000 <= keep_max_num_evs < 200000
        print(f"Slicing max {keep_max_num_evs} num evs.")

    def __call__(self, x):
        s = _as_stream(x)
        if len(s) > self.keep_max_N_evs:
            start = random.choice(range(len(s) - self.keep_max_N_evs + 1))
            s.beg += start
            s.end = s.beg + self.keep_max_N_evs
        return s


class RandomTimeFlip:
    """datasets.py:598-609."""

    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, x):
        s = _as_stream(x)
        if np.random.random() < self.p:
            assert not (s.aug["flip_x"][0] or s.aug["do_filter"][0]), "order: time flip comes first"
            s.aug["time_flip"] ^= 1
        return s


class Aug_FlipEvsAlongX:
    """datasets.py:501-521."""

    def __init__(self, H=None, W=None, p=0.5):
        if H is not None:
            assert 100 <= H <= 640
        if W is not None:
            assert 100 <= W <= 640
        assert 0.0 <= p <= 1.0
        self.H, self.W, self.p = H, W, p

    def __call__(self, x):
        s = _as_stream(x)
        W = self.W
        if W is None:
            W = int(np.float64(s.max_xy()[0]).astype(np.int64)) + 1
        if np.random.random() < self.p:
            s.aug["flip_x"] = 1
            s.aug["flip_w"] = W
        return s


class Aug_RandomShiftEvs:
    """datasets.py:524-549."""

    def __init__(self, H=None, W=None, max_shift=20):
        if H is not None:
            assert 100 <= H <= 640
        if W is not None:
            assert 100 <= W <= 640
        assert 0 <= max_shift <= 200
        self.H, self.W, self.max_shift = H, W, max_shift

    def __call__(self, x):
        s = _as_stream(x)
        H, W = self.H, self.W
        if W is None or H is None:
            mx, my = s.max_xy()
            if W is None:
                W = int(np.float64(mx).astype(np.int64)) + 1
            if H is None:
                H = int(np.float64(my).astype(np.int64)) + 1
        xs, ys = np.random.randint(-self.max_shift, self.max_shift + 1, size=(2,))
        s.aug["shift_x"], s.aug["shift_y"] = xs, ys
        s.aug["do_filter"], s.aug["filt_w"], s.aug["filt_h"] = 1, W, H
        return s


class EventArrToImg:
    """datasets.py:552-595 -> (H, W, 3) uint8 ndarray [pos, tss, neg] (fused HIP pass)."""

    def __init__(self, H=None, W=None, timeSurface=False):
        if H is not None:
            assert 100 <= H <= 640
        if W is not None:
            assert 100 <= W <= 640
        self.H, self.W, self.timeSurface = H, W, timeSurface
        if self.timeSurface:
            print("Using Time Surface!")

    def device_chw(self, x):
        s = _as_stream(x)
        H, W = self.H, self.W
        if W is None or H is None:
            mx, my = s.max_xy()
            if W is None:
                W = int(np.float64(mx).astype(np.int64)) + 1
            if H is None:
                H = int(np.float64(my).astype(np.int64)) + 1
        off, aug = s._dev()
        return rasterize(s.ev, off, H, W, self.timeSurface, aug)[0]

    def __call__(self, x):
        return self.device_chw(x).permute(1, 2, 0).contiguous().cpu().numpy()


class ToTensor:
    """torchvision.transforms.ToTensor for a uint8 HWC ndarray (datasets.py:637): CHW f32 / 255."""

    def __call__(self, img):
        return torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).to(torch.float32).div(255)


class Compose:
    def __init__(self, ts):
        self.transforms = list(ts)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x

    def __repr__(self):
        return "Compose(" + ", ".join(type(t).__name__ for t in self.transforms) + ")"


def build_transformNPY(is_train, args):
    """datasets.py:611-660.  Not in this round (torchvision arithmetic):
    Resize for data-dependent canvases, EventRandAugment, ColorJitter -- requested combinations
    raise NotImplementedError instead of silently differing."""
    t = []
    H, W = None, None
    if "imagenet" in args.data_path:
        H, W = args.input_H, args.input_W
        t += [ReshapeScaleXandY(newH=H, newW=W, oldH=480, oldW=640, is_train=is_train)]
        if is_train:
            H, W = int(480 * (256 / 480)), int(640 * (256 / 480))
    elif any(k in args.data_path for k in ("SS_final", "dsec", "DSEC")):
        H, W = 440, 640
    elif getattr(args, "fixed_canvas", False):
        H, W = args.input_H, args.input_W
    t += [SliceRandomMaxEvs(args.slice_max_evs)]
    if is_train:
        t += [RandomTimeFlip(), Aug_FlipEvsAlongX(H=H, W=W),
              Aug_RandomShiftEvs(H=H, W=W, max_shift=args.max_random_shift_evs)]
    t += [EventArrToImg(H, W, args.timesurface), ToTensor()]
    if (H, W) != (args.input_H, args.input_W):
        raise NotImplementedError("Resize/RandomCrop to the model size is torchvision arithmetic "
                                  "(SURVEY.md 8 f2): only canvases equal to (input_H, input_W) this round")
    t.append(T.EventChain(timesurface=args.timesurface, hotpixfilter=args.hotpixfilter,
                          num_stds=args.hotpix_num_stds, logtrafo=args.logtrafo, gammatrafo=args.gammatrafo,
                          gamma=args.gamma, normalize=args.normalize_events))
    if is_train and args.rand_aug:
        raise NotImplementedError("EventRandAugment is torchvision arithmetic (SURVEY.md 8 f2); pass --rand_aug 0")
    return Compose(t)


class DataAugmentationForPT:
    """datasets.py:26-82: returns (patches, visual_tokens (same tensor), mask)."""

    def __init__(self, args, is_train=True):
        if getattr(args, "color_jitter", 0):
            raise NotImplementedError("ColorJitter is torchvision arithmetic (SURVEY.md 8 f2); pass --color_jitter 0")
        if args.discrete_vae_type != "event":
            raise NotImplementedError()
        self.common_transform = Compose([build_transformNPY(is_train, args), T.CreateTwoPic()])
        if args.masking == "random":
            self.masked_position_generator = MaskingGeneratorRandomLocation(
                args.window_size, num_masking_patches=args.num_mask_patches)
        elif args.masking == "block":
            self.masked_position_generator = MaskingGenerator(
                args.window_size, num_masking_patches=args.num_mask_patches,
                max_num_patches=args.max_mask_patches_per_block,
                min_num_patches=args.min_mask_patches_per_block)
        else:
            raise ValueError(f"Need to chose proper masking scheme. {args.masking} does not exist.")

    def __call__(self, image):
        for_patches, for_visual_tokens = self.common_transform(image)
        return for_patches, for_visual_tokens, self.masked_position_generator()

    def __repr__(self):
        return ("(DataAugmentationForPT,\n  common_transform = %s,\n  Masked position generator = %s,\n)"
                % (self.common_transform, self.masked_position_generator))


class SyntheticEventDataset(torch.utils.data.Dataset):
    """Seeded synthetic (N,4) event streams with the layout of dataset_folder.py:275-302 -- the
    stand-in for npyFolder when no dataset is on disk (all BASELINE configs are synthetic)."""

    def __init__(self, n_samples, n_events, H, W, transform=None, seed=1234):
        self.n, self.ne, self.H, self.W, self.transform, self.seed = n_samples, n_events, H, W, transform, seed

    def __len__(self):
        return self.n

    def events(self, i):
        g = np.random.default_rng(self.seed + i)
        n = self.ne
        return np.stack([g.integers(0, self.W, n), g.integers(0, self.H, n),
                         np.sort(g.integers(0, 300000, n)), g.integers(0, 2, n) * 2 - 1], 1).astype(np.float64)

    def __getitem__(self, i):
        x = self.events(i)
        return (self.transform(x) if self.transform is not None else x), 0


def build_pretraining_dataset(args, is_train=True):
    """datasets.py:146-174.  Folder walking / .npy loading is I/O plumbing outside the hot path
    (SURVEY.md 2.1 row 8); ``--data_path synthetic`` builds the seeded synthetic stand-in."""
    if args.data_path != "synthetic":
        raise NotImplementedError("only --data_path synthetic this round (dataset folder I/O is out of scope)")
    args.fixed_canvas = True                  # synthetic streams are generated on the model's own canvas
    transform = DataAugmentationForPT(args, is_train)
    print("Data Aug = %s" % str(transform))
    n = getattr(args, "synthetic_samples", 64)
    return SyntheticEventDataset(n if is_train else max(2, n // 8), args.slice_max_evs, args.input_H, args.input_W,
                                 transform=transform, seed=1234 if is_train else 4321)


class EventBatchPipeline:
    """Batched on-GPU form of the chain above for fixed canvases: CSR events in HBM ->
    fused augment+rasterize -> fused event_norm -> f32 [B, C, H, W], plus masks.
    Host work per batch = the random draws and one 56-byte record per sample."""

    def __init__(self, H, W, out_chans=2, time_surface=False, train_augs=False, max_shift=8,
                 hotpix=True, num_stds=10.0, normalize=True):
        self.H, self.W, self.out_chans, self.ts = H, W, out_chans, time_surface
        self.train_augs, self.max_shift = train_augs, max_shift
        self.flags = ((0 if time_surface else T.EV_RM_TS) | (T.EV_HOTPIX if hotpix else 0)
                      | (T.EV_NORMALIZE if normalize else 0))
        self.num_stds = num_stds

    def draw_augs(self, B):
        a = _new_aug_array(B)
        if self.train_augs:
            for b in range(B):       # same generators / order as the reference chain, per sample
                if np.random.random() < 0.5:
                    a["time_flip"][b] = 1
                if np.random.random() < 0.5:
                    a["flip_x"][b] = 1
                    a["flip_w"][b] = self.W
                xs, ys = np.random.randint(-self.max_shift, self.max_shift + 1, size=(2,))
                a["shift_x"][b], a["shift_y"][b] = xs, ys
                a["do_filter"][b], a["filt_w"][b], a["filt_h"][b] = 1, self.W, self.H
        return a

    def __call__(self, ev, offsets, aug_host=None):
        B = offsets.numel() - 1
        aug = None
        if aug_host is None and self.train_augs:
            aug_host = self.draw_augs(B)
        if aug_host is not None:
            aug = torch.from_numpy(aug_host.view(np.uint8)).to(ev.device, non_blocking=True)
        img = rasterize(ev, offsets, self.H, self.W, self.ts, aug, strict=False)
        return T.event_norm(img, self.flags, self.num_stds, 0.5, self.out_chans)
