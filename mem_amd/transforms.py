"""Tensor-level event transforms -- mirror of /root/reference/mem/transforms.py:66-72,200-275,
332-348 (CreateTwoPic, LogTransform, GammaTransform, NormalizeEvent, RemoveTimesurface,
RemoveHotPixels, ToUnit8, ToFloat32).

Each class keeps the reference's constructor and ``__call__(x[3,H,W] f32) -> tensor``; the
arithmetic runs in the fused HIP kernel csrc/event_norm.hip (one launch applies any subset
in the reference's fixed order, datasets.py:644-653).  Tensors that arrive on the host are
moved to the GPU and the result is returned on the input's device.  EventRandAugment /
RandomResizedCrop* (torchvision arithmetic) are SURVEY.md section 8 row f2: mem_amd/augment.py.
"""
import torch

from ._lib import check, declare, f32, i32, lib, ptr, require_gpu, stream_ptr, vp

declare({"memhip_event_norm": (i32, [vp, i32, i32, i32, i32, i32, f32, f32, vp, i32, vp]),
         "memhip_event_norm_topk": (i32, [vp, i32, i32, i32, i32, i32, i32, f32, vp, i32, vp, vp])})

EV_RM_TS, EV_HOTPIX, EV_LOG, EV_GAMMA, EV_NORMALIZE = 1, 2, 4, 8, 16


def event_norm(x, flags, num_stds=10.0, gamma=0.5, out_chans=3, num_hot_pixels=None):
    """x: u8 or f32 [B,3,H,W] on the GPU -> f32 [B,out_chans,H,W].  num_hot_pixels: the top-k form of RemoveHotPixels
    (transforms.py:257-263) instead of the mean + num_stds * std threshold."""
    require_gpu()
    assert x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.is_contiguous()
    assert x.dtype in (torch.uint8, torch.float32)
    B, _, H, W = x.shape
    out = torch.empty((B, out_chans, H, W), dtype=torch.float32, device=x.device)
    if num_hot_pixels is not None:
        keys = torch.empty((B,), dtype=torch.int64, device=x.device)
        check(lib.memhip_event_norm_topk(ptr(x), int(x.dtype == torch.uint8), B, H, W, int(flags) | EV_HOTPIX,
                                         int(num_hot_pixels), float(gamma), ptr(out), out_chans, ptr(keys), stream_ptr()),
              "event_norm_topk")
        return out
    check(lib.memhip_event_norm(ptr(x), int(x.dtype == torch.uint8), B, H, W, int(flags), float(num_stds),
                                float(gamma), ptr(out), out_chans, stream_ptr()), "event_norm")
    return out


def _run(x, flags, num_stds=10.0, gamma=0.5, num_hot_pixels=None):
    dev = x.device
    y = event_norm(x.to("cuda", dtype=torch.float32).contiguous()[None], flags, num_stds, gamma, 3, num_hot_pixels)[0]
    return y.to(dev)


class CreateTwoPic:
    def __call__(self, img):
        return img, img


class RemoveTimesurface:
    def __call__(self, x):
        return _run(x, EV_RM_TS)


class RemoveHotPixels:
    """transforms.py:249-275.  num_hot_pixels selects the top-k branch (:257-263); equal values at the selection boundary
    are ordered by flat index there (the reference leaves them to torch.argsort(stable=False))."""

    def __init__(self, num_stds=10, num_hot_pixels=None):
        self.num_stds = num_stds
        self.num_hot_pixels = num_hot_pixels

    def __call__(self, x):
        return _run(x, EV_HOTPIX, num_stds=self.num_stds, num_hot_pixels=self.num_hot_pixels)


class LogTransform:
    def __call__(self, x):
        return _run(x, EV_LOG)


class GammaTransform:
    def __init__(self, gamma=0.5):
        self.gamma = gamma

    def __call__(self, x):
        return _run(x, EV_GAMMA, gamma=self.gamma)


class NormalizeEvent:
    def __call__(self, x):
        return _run(x, EV_NORMALIZE)


class EventChain:
    """RemoveTimesurface -> RemoveHotPixels -> Log -> Gamma -> NormalizeEvent in ONE launch
    (the order build_transformNPY fixes, datasets.py:644-653)."""

    def __init__(self, timesurface=0, hotpixfilter=1, num_stds=10, logtrafo=0, gammatrafo=0, gamma=0.5,
                 normalize=1):
        self.flags = ((0 if timesurface else EV_RM_TS) | (EV_HOTPIX if hotpixfilter else 0)
                      | (EV_LOG if logtrafo else 0) | (EV_GAMMA if gammatrafo else 0)
                      | (EV_NORMALIZE if normalize else 0))
        self.num_stds, self.gamma = num_stds, gamma

    def __call__(self, x):
        return _run(x, self.flags, self.num_stds, self.gamma)


class ToUnit8:
    """transforms.py:341-348."""

    def __call__(self, x):
        return (255 * x).to(torch.uint8)


class ToFloat32:
    """transforms.py:332-339."""

    def __call__(self, x):
        return x.to(torch.float32) / 255
