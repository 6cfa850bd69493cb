"""Torch/CPU restatement of the tensor-level event transforms.  TEST INFRASTRUCTURE.

Follows /root/reference/mem/transforms.py:200-275,332-348 in the order fixed by
build_transformNPY (/root/reference/mem/datasets.py:644-658).  Inputs are
``[3, H, W] float32`` tensors ``[pos, tss, neg]``; all functions return a new
tensor.  Pinned by oracle/gen_golden.py -> tests/golden/transforms_*.npz.
"""
import torch


def remove_timesurface(x):
    """transforms.py:239-247: channel 1 <- 0."""
    x = x.clone()
    x[1] = 0.0
    return x.float()


def remove_hot_pixels(x, num_stds=10.0):
    """transforms.py:249-275 (num_hot_pixels=None branch).

    thr = mean + num_stds * std over channels {0,2} (torch.std: unbiased, n-1);
    every pixel (y,x) where EITHER polarity channel exceeds thr is zeroed in
    BOTH polarity channels (:273-274: the flat index over the [2,H,W] slice is
    unravelled against the 3-channel shape -- the y,x components are unaffected
    by that, only the channel component is, and the channel is not used)."""
    x = x.clone()
    pol = x[0::2]
    thr = pol.mean() + num_stds * pol.std()
    hot = (pol > thr).any(dim=0)
    x[0][hot] = 0
    x[2][hot] = 0
    return x


def remove_hot_pixels_topk(x, num_hot_pixels):
    """transforms.py:257-263,270-274 (num_hot_pixels branch), restated with the reference's own operations: the clamp
    to sum / 4, torch.argsort of the flattened [pos, neg] planes, the last int(k) indices, unravelled to (y, x), both
    polarities zeroed.  torch.argsort(stable=False) orders EQUAL values as the sort implementation pleases, so the
    result is defined by the reference only when the k-th and (k+1)-th largest values differ (``topk_is_tie_free``)."""
    x = x.clone()
    pol = x[0::2]
    flat = pol.flatten()
    k = num_hot_pixels
    if k >= pol.sum() / 4:
        k = pol.sum() / 4
    idx = torch.atleast_1d(torch.argsort(flat)[len(flat) - int(k):])
    hw = x.shape[1] * x.shape[2]
    yy, xx = (idx % hw) // x.shape[2], idx % x.shape[2]
    x[0, yy, xx] = 0
    x[2, yy, xx] = 0
    return x


def topk_clamped(x, num_hot_pixels):
    pol = x[0::2]
    k = num_hot_pixels
    if k >= pol.sum() / 4:
        k = pol.sum() / 4
    return int(k)


def topk_is_tie_free(x, num_hot_pixels):
    k = topk_clamped(x, num_hot_pixels)
    v = torch.sort(x[0::2].flatten()).values
    n = len(v)
    return k <= 0 or k >= n or bool(v[n - k] > v[n - k - 1])


def normalize_event(x):
    """transforms.py:225-237: divide channels {0,2} by their joint max when it
    is non-zero (multiplication by the fp32 reciprocal, as the reference does)."""
    x = x.clone()
    mx = x[0::2].max()
    if mx != 0:
        factor = 1.0 / mx
        x[0::2] = x[0::2] * factor
    return x.float()


def log_transform(x):
    """transforms.py:200-210: log(1+x) on channels 0 and 2."""
    x = x.clone()
    x[0] = torch.log(x[0] + 1.0)
    x[2] = torch.log(x[2] + 1.0)
    return x.float()


def gamma_transform(x, gamma=0.5):
    """transforms.py:212-222: x**gamma on channels 0 and 2."""
    x = x.clone()
    x[0] = x[0] ** gamma
    x[2] = x[2] ** gamma
    return x.float()


def to_uint8(x):
    """transforms.py:341-348: (255*x).to(uint8) (truncation)."""
    return (255 * x).to(torch.uint8)


def to_float32(x):
    """transforms.py:332-339: uint8 -> float32 / 255."""
    return x.to(torch.float32) / 255


def event_chain(x, timesurface=False, hotpixfilter=True, num_stds=10.0, logtrafo=False,
                gammatrafo=False, gamma=0.5, normalize=True):
    """The fixed order of datasets.py:644-653 (after ToTensor / Resize / Crop)."""
    if not timesurface:
        x = remove_timesurface(x)
    if hotpixfilter:
        x = remove_hot_pixels(x, num_stds)
    if logtrafo:
        x = log_transform(x)
    if gammatrafo:
        x = gamma_transform(x, gamma)
    if normalize:
        x = normalize_event(x)
    return x
