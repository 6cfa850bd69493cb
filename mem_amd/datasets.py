"""Event data path -- mirror of /root/reference/mem/datasets.py (event classes :464-660,
DataAugmentationForPT :26-82, build_pretraining_dataset :146-174).

MI355X design: the reference runs five NumPy passes per sample in DataLoader workers
(slice -> time flip -> x flip -> shift+filter -> np.add.at).  Here the event-level classes
only *record* their parameters on a lazy ``EventStream`` (events live in HBM as the
reference's (N,4) float64 rows); ``EventArrToImg`` then runs ONE fused HIP pass
(csrc/raster.hip) that applies the whole chain while it reads each event once.  The random
draws are made on the host with the same generators, in the same order, as the reference
(``random.choice``, ``np.random.random``, ``np.random.randint``), so results are equal draw
for draw.  ``augment.BatchAugPipeline`` is the batched form training uses (raw events + per-sample draw records are collated
and the whole chain runs on the GPU once per batch); ``EventBatchPipeline`` is its fixed-canvas special case that
bench.py times.
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
_BINNED_MAX_PIXELS = 64 * 40000



class EventAug(C.Structure):
    """== memhip_event_aug_t (include/memhip.h)."""
    _fields_ = [("scale_x", C.c_double), ("scale_y", C.c_double), ("time_flip", C.c_int32),
                ("flip_x", C.c_int32), ("flip_w", C.c_int64), ("shift_x", C.c_int32),
                ("shift_y", C.c_int32), ("do_filter", C.c_int32), ("filt_w", C.c_int32),
                ("filt_h", C.c_int32), ("infer", C.c_int32)]

    def __init__(self):
        super().__init__()
        self.scale_x = self.scale_y = 1.0


AUG_DTYPE = np.dtype([("scale_x", "<f8"), ("scale_y", "<f8"), ("time_flip", "<i4"), ("flip_x", "<i4"),
                      ("flip_w", "<i8"), ("shift_x", "<i4"), ("shift_y", "<i4"), ("do_filter", "<i4"),
                      ("filt_w", "<i4"), ("filt_h", "<i4"), ("infer", "<i4")])
assert AUG_DTYPE.itemsize == C.sizeof(EventAug) == 56


def _new_aug_array(n):
    a = np.zeros(n, dtype=AUG_DTYPE)
    a["scale_x"] = 1.0
    a["scale_y"] = 1.0
    return a


def rasterize(ev, offsets, H, W, time_surface=False, aug=None, strict=True, binned=None, status_out=None):
    """ev f64 [n,4] (cuda), offsets i64 [B+1] (cuda), aug = uint8 cuda view of B aug records or
    None -> u8 [B,3,H,W] (cuda).  strict: raise IndexError like the reference when an event lands
    outside the canvas (costs one host sync); with strict=False pass a list as status_out to receive the
    per-sample i32 count of such events (device tensor) and check it later.  binned: None = choose by size,
    True/False = force the two-pass long-stream kernels / the single-pass kernels."""
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
    if status_out is not None:
        status_out.append(status)
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


class TransformNPY:
    """What build_transformNPY returns (datasets.py:611-660): the per-sample chain events (N,4) -> f32 [3,H,W].

    ``__call__`` keeps the reference's per-sample surface (it runs the batched GPU chain with B = 1 and returns a CPU
    tensor like the reference's transform does); training does NOT go through it: the dataset hands out raw events +
    ``draw()`` records and the whole batch runs through ``augment.BatchAugPipeline`` once per step (collate_raw)."""

    def __init__(self, is_train, args, with_jitter=False, out_chans=3):
        from .augment import BatchAugPipeline, ChainConfig
        self.cfg = ChainConfig(args, is_train)
        self.cfg.apply_jitter = with_jitter
        self.pipe = BatchAugPipeline(self.cfg, out_chans)
        if self.cfg.slice_max:
            assert 5000 <= self.cfg.slice_max < 200000
            print(f"Slicing max {self.cfg.slice_max} num evs.")
        if self.cfg.time_surface:
            print("Using Time Surface!")

    def draw(self, n_events):
        from .augment import draw_sample
        return draw_sample(self.cfg, n_events)

    def __call__(self, x):
        ev = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda() if isinstance(x, np.ndarray) else x.cuda()
        d = self.draw(ev.shape[0])
        out, st = self.pipe(ev, [0, ev.shape[0]], [d], return_stages=True)
        code = int(st["status"][0].item())
        if code & (1 << 28):
            raise IndexError("event outside the canvas (reference: np.add.at IndexError)")
        if code != 0:
            raise ValueError("empty sample or canvas beyond the sensor bound (reference: max() of an empty array)")
        return out[0].cpu()

    def __repr__(self):
        c = self.cfg
        return (f"TransformNPY(canvas={c.canvas or 'from events'}, scale={c.scale}, resize={c.resize}, crop={c.crop}, "
                f"flags={c.flags}, rand_aug={c.rand_aug}, color_jitter={c.color_jitter if c.apply_jitter else None})")


def build_transformNPY(is_train, args):
    """datasets.py:611-660."""
    return TransformNPY(is_train, args)


class DataAugmentationForPT:
    """datasets.py:26-82: returns (patches, visual_tokens (same tensor), mask)."""

    def __init__(self, args, is_train=True):
        if args.discrete_vae_type != "event":
            raise NotImplementedError()
        if getattr(args, "data_set", "npy") == "dsec_semseg":
            raise NotImplementedError("build_transform_dsec (segmentation) is outside the pretraining path")
        # build_transformNPY + ColorJitter(color_jitter, 0, color_jitter) + CreateTwoPic (:33-37)
        self.common_transform = TransformNPY(is_train, args, with_jitter=True)
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

    def draw(self, n_events):
        """(transform draws, mask) in the reference's per-sample order: transform chain first, mask last (:71-75)."""
        return self.common_transform.draw(n_events), self.masked_position_generator()

    def __call__(self, image):
        x = self.common_transform(image)
        return x, x, self.masked_position_generator()

    def __repr__(self):
        return ("(DataAugmentationForPT,\n  common_transform = %s,\n  Masked position generator = %s,\n)"
                % (self.common_transform, self.masked_position_generator))


class RawEventDataset(torch.utils.data.Dataset):
    """(events, draws, mask) per sample: CPU only (DataLoader workers never touch the GPU).  ``source(i)`` -> (N,4)
    float64 ndarray; ``aug`` = DataAugmentationForPT.  The slice window of SliceRandomMaxEvs is applied here (a view),
    so the collated batch is one contiguous CSR buffer."""

    def __init__(self, n, source, aug):
        self.n, self.source, self.aug = n, source, aug

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        ev = self.source(i)
        d, mask = self.aug.draw(len(ev))
        ev = ev[d.beg:d.end]
        d.beg, d.end = 0, len(ev)
        return (ev, d, mask), 0

    def collate(self, batch):
        evs = [b[0][0] for b in batch]
        lens = np.array([0] + [len(e) for e in evs], dtype=np.int64)
        return {"events": torch.from_numpy(np.ascontiguousarray(np.concatenate(evs, 0), dtype=np.float64)),
                "offsets": np.cumsum(lens), "draws": [b[0][1] for b in batch],
                "masks": np.stack([b[0][2] for b in batch]), "pipe": self.aug.common_transform.pipe}, \
            torch.zeros(len(batch), dtype=torch.int64)


# sensor geometry of the synthetic stand-ins, by data_path keyword (W, H)
_SENSOR = {"caltech": (240, 180), "Caltech": (240, 180), "ncars": (120, 100), "N-Cars": (120, 100), "imagenet": (640, 480),
           "dsec": (640, 480), "DSEC": (640, 480), "SS_final": (640, 480)}


class SyntheticEventSource:
    """Seeded synthetic (N,4) event streams with the layout of dataset_folder.py:275-302.  With ``vary_extent`` every
    sample occupies its own sub-rectangle of the sensor (like N-Caltech101 recordings, whose canvas the reference
    infers per sample from the data)."""

    def __init__(self, n_events, W, H, seed=1234, vary_extent=False):
        self.ne, self.W, self.H, self.seed, self.vary = n_events, W, H, seed, vary_extent

    def __call__(self, i):
        g = np.random.default_rng(self.seed + i)
        W, H = self.W, self.H
        if self.vary:
            W, H = int(g.integers(max(100, (2 * W) // 3), W + 1)), int(g.integers(max(100, (2 * H) // 3), H + 1))
        n = self.ne
        return np.stack([g.integers(0, W, n), g.integers(0, H, n), np.sort(g.integers(0, 300000, n)),
                         g.integers(0, 2, n) * 2 - 1], 1).astype(np.float64)


class SyntheticEventDataset(torch.utils.data.Dataset):
    """Per-sample form (transform applied in __getitem__ through the B = 1 compat path; tests / small tools)."""

    def __init__(self, n_samples, n_events, H, W, transform=None, seed=1234):
        self.n, self.transform = n_samples, transform
        self.src = SyntheticEventSource(n_events, W, H, seed)

    def __len__(self):
        return self.n

    def events(self, i):
        return self.src(i)

    def __getitem__(self, i):
        x = self.events(i)
        return (self.transform(x) if self.transform is not None else x), 0


class NpyFolderSource:
    """root/<class>/<file>.npy|npz in sorted order (dataset_folder.py:npyFolder) with the reference's loader choice
    (datasets.py:158-172); events are loaded to host memory (the conversion kernels of process_data run per batch)."""

    def __init__(self, root, loader):
        import os
        self.files = []
        for cls in sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))):
            for f in sorted(os.listdir(os.path.join(root, cls))):
                if f.endswith((".npy", ".npz")):
                    self.files.append(os.path.join(root, cls, f))
        self.loader = loader

    def __len__(self):
        return len(self.files)

    def __call__(self, i):
        return self.loader(self.files[i])


def _host_loader(args):
    """CPU forms of the loaders (DataLoader workers): (N,4) float64 ndarrays, same arithmetic as dataset_folder.py:275-302
    (the GPU forms live in process_data.py)."""
    dp = args.data_path
    if getattr(args, "data_set", "npy") == "dsec_semseg":
        def dsec(path):
            data = np.load(path).astype(float)
            data[:, 3] = 2 * data[:, 3] - 1
            return data[data[:, 1] < 440]
        return dsec
    if "imagenet" in dp and "npy" in dp:
        def imgnet(path):
            data = np.load(path)
            ps = data["p"].astype(np.int8) * 2 - 1
            return np.vstack([data["x"], data["y"], data["t"], ps]).T.astype(float)
        return imgnet
    return lambda path: np.load(path)


def build_pretraining_dataset(args, is_train=True):
    """datasets.py:146-174.  A data_path that exists is read like the reference reads it (root/train|val/<class>/*.npy);
    a path that does not exist raises like the reference's assert, unless it is the literal ``synthetic`` or the caller
    opted in with ``--synthetic_if_missing 1``: then seeded synthetic streams of the geometry the data_path names
    (N-Caltech101: per-sample extents inside 240 x 180) stand in, with a WARNING in every case."""
    import os
    transform = DataAugmentationForPT(args, is_train)
    print("Data Aug = %s" % str(transform))
    root = None
    for a, b in (("train", "val"), ("extracted_train", "extracted_val"), ("train_events", "test_events")):
        r = os.path.join(args.data_path, a if is_train else b)
        if os.path.exists(r):
            root = r
            break
    if root is not None:
        src = NpyFolderSource(root, _host_loader(args))
        return RawEventDataset(len(src), src, transform)
    if args.data_path != "synthetic":
        # the reference asserts that the root exists (datasets.py:149-154): a missing path fails unless the caller opted in
        if not getattr(args, "synthetic_if_missing", 0):
            raise AssertionError(f"{args.data_path} not found (pass --data_path synthetic, or --synthetic_if_missing 1 "
                                 f"to train on seeded synthetic event streams instead)")
    n = getattr(args, "synthetic_samples", 64)
    n = n if is_train else max(2, n // 8)
    cfg = transform.common_transform.cfg
    key = next((k for k in _SENSOR if k in args.data_path), None)
    if key is None:                                   # "synthetic" (or an unrecognised name): streams on the model's own canvas
        W, H, vary = args.input_W, args.input_H, False
        cfg.canvas = cfg.canvas or (args.input_H, args.input_W)
    else:
        (W, H), vary = _SENSOR[key], cfg.canvas is None
        cfg.canvas_max = (max(cfg.canvas_max[0], H), max(cfg.canvas_max[1], W))
    if args.data_path != "synthetic":
        print(f"WARNING: {args.data_path} does not exist here -- --synthetic_if_missing 1: training on SEEDED SYNTHETIC event "
              f"streams ({key or 'model canvas'} geometry, {W}x{H}), not on data", flush=True)
    src = SyntheticEventSource(args.slice_max_evs, W, H, seed=1234 if is_train else 4321, vary_extent=vary)
    return RawEventDataset(n, src, transform)


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
