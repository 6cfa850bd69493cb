"""ORACLE (test infrastructure, not product): torch/CPU restatement of the image-space augmentation chain that the
reference builds from torchvision tensor ops.

PARITY UNPINNED (third party): torchvision is not vendored under /root/reference and not importable in the build
container, so these functions restate its PUBLISHED tensor algorithms (torchvision.transforms.functional /
_functional_tensor, 0.13-0.20 are identical for the calls below) and are anchored on the reference's call sites:
    mem/datasets.py:36      transforms.ColorJitter(args.color_jitter, 0, args.color_jitter)
    mem/datasets.py:637     transforms.ToTensor()
    mem/datasets.py:639     transforms.Resize((H, W), InterpolationMode.BILINEAR, antialias=True)
    mem/datasets.py:642     transforms.RandomCrop((H, W), pad_if_needed=True)
    mem/transforms.py:292-330   _apply_op (F.affine / F.rotate / adjust_* / posterize / solarize / autocontrast / equalize)
    mem/transforms.py:349-484   EventRandAugment (op list, magnitude bins, draw order) -- this part IS reference code
                                and is restated 1:1 (draws: torch.randint(len), torch.randint(magnitude + 1), torch.randint(2))
Arithmetic that torch itself provides (F.interpolate(antialias=True), F.grid_sample, conv2d) is CALLED, not restated:
the installed torch is the CPU oracle arithmetic.
"""
import math

import torch
import torch.nn.functional as F

OPS = ["Identity", "ShearX", "ShearY", "TranslateX", "TranslateY", "Rotate", "Brightness", "Color", "Contrast",
       "Sharpness", "Posterize", "Solarize", "AutoContrast", "Equalize"]          # transforms.py:391-392 (small=False)


# ------------------------------------------------------------------ ToTensor / Resize / RandomCrop
def to_tensor_u8_hwc(img_hwc_u8):
    """torchvision ToTensor on a uint8 HWC ndarray: CHW float32 / 255."""
    return torch.from_numpy(img_hwc_u8).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def resize_bilinear_aa(x, size):
    """F.resize(tensor, size, BILINEAR, antialias=True) == interpolate(mode='bilinear', align_corners=False, antialias=True)."""
    if tuple(x.shape[-2:]) == tuple(size):
        return x
    return F.interpolate(x[None], size=tuple(size), mode="bilinear", align_corners=False, antialias=True)[0]


def random_crop_params(h, w, th, tw, gen=None):
    """RandomCrop.get_params on the (padded) image: no draw when the size already matches, else i then j."""
    h, w = padded_size(h, w, th, tw)
    if w == tw and h == th:
        return 0, 0
    i = int(torch.randint(0, h - th + 1, size=(1,), generator=gen).item())
    j = int(torch.randint(0, w - tw + 1, size=(1,), generator=gen).item())
    return i, j


def padded_size(h, w, th, tw):
    """RandomCrop(pad_if_needed): a deficit d pads BOTH sides by d (F.pad(img, [d, 0]) = left/right d)."""
    return (h + 2 * (th - h) if h < th else h), (w + 2 * (tw - w) if w < tw else w)


def random_crop(x, th, tw, i, j):
    _, h, w = x.shape
    if w < tw:
        x = F.pad(x, [tw - w, tw - w, 0, 0])
    if h < th:
        x = F.pad(x, [0, 0, th - h, th - h])
    return x[:, i:i + th, j:j + tw]


# ------------------------------------------------------------------ _functional_tensor pieces
def _bound(x):
    return 1.0 if x.is_floating_point() else 255.0


def _blend(img1, img2, ratio):
    ratio = float(ratio)
    return (ratio * img1 + (1.0 - ratio) * img2).clamp(0, _bound(img1)).to(img1.dtype)


def rgb_to_grayscale(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(dim=-3)


def adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def adjust_saturation(img, f):
    return _blend(img, rgb_to_grayscale(img), f)


def adjust_contrast(img, f):
    dtype = img.dtype if img.is_floating_point() else torch.float32
    mean = torch.mean(rgb_to_grayscale(img).to(dtype), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, f)


def _cast_round_u8(x, out_dtype):
    if out_dtype == torch.uint8:
        x = torch.round(x)
    return x.to(out_dtype)


def adjust_sharpness(img, f):
    if img.size(-1) <= 2 or img.size(-2) <= 2:
        return img
    dtype = img.dtype if img.is_floating_point() else torch.float32
    kernel = torch.ones((3, 3), dtype=dtype)
    kernel[1, 1] = 5.0
    kernel /= kernel.sum()
    kernel = kernel.expand(img.shape[-3], 1, 3, 3)
    tmp = F.conv2d(img.to(dtype)[None], kernel, groups=img.shape[-3])[0]
    tmp = _cast_round_u8(tmp, img.dtype)
    degenerate = img.clone()
    degenerate[..., 1:-1, 1:-1] = tmp
    return _blend(img, degenerate, f)


def posterize(img, bits):
    mask = (256 - 2 ** (8 - int(bits))) & 255                    # == torch.tensor(-int(2 ** (8 - bits)), dtype=uint8)
    return img & torch.tensor(mask, dtype=torch.uint8)


def solarize(img, threshold):
    return torch.where(img >= threshold, 255 - img, img)


def autocontrast(img):
    bound = 255.0
    minimum = img.amin(dim=(-2, -1), keepdim=True).to(torch.float32)
    maximum = img.amax(dim=(-2, -1), keepdim=True).to(torch.float32)
    scale = bound / (maximum - minimum)
    eq = torch.isfinite(scale).logical_not()
    minimum[eq] = 0
    scale[eq] = 1
    return ((img - minimum) * scale).clamp(0, bound).to(img.dtype)


def _scale_channel(ch):
    hist = torch.bincount(ch.reshape(-1).to(torch.int64), minlength=256).to(torch.float32)
    nonzero = hist[hist != 0]
    step = torch.div(nonzero[:-1].sum(), 255, rounding_mode="floor")
    if step == 0:
        return ch
    lut = torch.div(torch.cumsum(hist, 0) + torch.div(step, 2, rounding_mode="floor"), step, rounding_mode="floor")
    lut = F.pad(lut, [1, 0])[:-1].clamp(0, 255)
    return lut[ch.to(torch.int64)].to(torch.uint8)


def equalize(img):
    return torch.stack([_scale_channel(img[c]) for c in range(img.shape[0])])


def inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision.transforms.functional._get_inverse_affine_matrix (inverted=True)."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _affine_grid(theta, w, h, ow, oh):
    d = 0.5
    base = torch.empty(1, oh, ow, 3, dtype=theta.dtype)
    base[..., 0].copy_(torch.linspace(-ow * 0.5 + d, ow * 0.5 + d - 1, steps=ow))
    base[..., 1].copy_(torch.linspace(-oh * 0.5 + d, oh * 0.5 + d - 1, steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=theta.dtype)
    return base.view(1, oh * ow, 3).bmm(rescaled).view(1, oh, ow, 2)


def _apply_matrix(img, matrix):
    """_FT.affine / _FT.rotate (expand=False) with BILINEAR interpolation and fill=None on a uint8 tensor."""
    h, w = img.shape[-2:]
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    grid = _affine_grid(theta, w, h, w, h)
    out = F.grid_sample(img.to(torch.float32)[None], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0]
    return _cast_round_u8(out, img.dtype)


def affine(img, angle, translate, scale, shear):
    """F.affine(tensor, center=None): centre [0, 0] in the grid's centred coordinates, translate as floats."""
    return _apply_matrix(img, inverse_affine_matrix([0.0, 0.0], angle, [1.0 * t for t in translate], scale, shear))


def rotate(img, angle):
    """F.rotate(tensor, angle, expand=False, center=None): matrix of -angle."""
    return _apply_matrix(img, inverse_affine_matrix([0.0, 0.0], -angle, [0.0, 0.0], 1.0, [0.0, 0.0]))


def op_matrix(op_name, magnitude):
    """The inverse affine matrix _apply_op (transforms.py:292-330) hands to torchvision for the geometric ops."""
    if op_name == "ShearX":
        return inverse_affine_matrix([0.0, 0.0], 0.0, [0.0, 0.0], 1.0, [math.degrees(magnitude), 0.0])
    if op_name == "ShearY":
        return inverse_affine_matrix([0.0, 0.0], 0.0, [0.0, 0.0], 1.0, [0.0, math.degrees(magnitude)])
    if op_name == "TranslateX":
        return inverse_affine_matrix([0.0, 0.0], 0.0, [1.0 * int(magnitude), 0.0], 1.0, [0.0, 0.0])
    if op_name == "TranslateY":
        return inverse_affine_matrix([0.0, 0.0], 0.0, [0.0, 1.0 * int(magnitude)], 1.0, [0.0, 0.0])
    if op_name == "Rotate":
        return inverse_affine_matrix([0.0, 0.0], -magnitude, [0.0, 0.0], 1.0, [0.0, 0.0])
    return None


def apply_op(img, op_name, magnitude):
    """mem/transforms.py:292-330 _apply_op (interpolation=BILINEAR, fill=None) on uint8 [3,H,W]."""
    m = op_matrix(op_name, magnitude)
    if m is not None:
        return _apply_matrix(img, m)
    if op_name == "Brightness":
        return adjust_brightness(img, 1.0 + magnitude)
    if op_name == "Color":
        return adjust_saturation(img, 1.0 + magnitude)
    if op_name == "Contrast":
        return adjust_contrast(img, 1.0 + magnitude)
    if op_name == "Sharpness":
        return adjust_sharpness(img, 1.0 + magnitude)
    if op_name == "Posterize":
        return posterize(img, int(magnitude))
    if op_name == "Solarize":
        return solarize(img, magnitude)
    if op_name == "AutoContrast":
        return autocontrast(img)
    if op_name == "Equalize":
        return equalize(img)
    if op_name == "Identity":
        return img
    raise ValueError(op_name)


# ------------------------------------------------------------------ EventRandAugment (reference code, transforms.py:349-484)
def augmentation_space(num_bins, image_size):
    """transforms.py:407-425."""
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


def rand_augment_draw(height, width, num_ops=2, magnitude=20, num_bins=31, gen=None):
    """The draws of EventRandAugment.forward (transforms.py:441-463), in its order: [(op_name, magnitude), ...]."""
    meta = augmentation_space(num_bins, (height, width))
    out = []
    for _ in range(num_ops):
        op_index = int(torch.randint(len(meta), (1,), generator=gen).item())
        name = list(meta.keys())[op_index]
        mags, signed = meta[name]
        r0 = torch.randint(magnitude + 1, (1,), generator=gen).item()
        r1 = torch.randint(2, (1,), generator=gen)
        m = float(mags[r0].item()) if mags.ndim > 0 else 0.0
        if signed and r1:
            m *= -1.0
        out.append((name, m))
    return out


def rand_augment(img_u8, draws):
    for name, m in draws:
        img_u8 = apply_op(img_u8, name, m)
    return img_u8


# ------------------------------------------------------------------ ColorJitter(b, 0, s)
def color_jitter_draw(b, s, gen=None):
    """torchvision ColorJitter.get_params with contrast = hue = None (value 0 -> None): randperm(4) is ALWAYS drawn,
    then a uniform_ for each non-None factor in the order brightness, saturation."""
    fn_idx = torch.randperm(4, generator=gen)
    bf = None if not b else float(torch.empty(1).uniform_(max(0.0, 1.0 - b), 1.0 + b, generator=gen))
    sf = None if not s else float(torch.empty(1).uniform_(max(0.0, 1.0 - s), 1.0 + s, generator=gen))
    return [int(v) for v in fn_idx], bf, sf


def color_jitter(x, fn_idx, bf, sf):
    for fn_id in fn_idx:
        if fn_id == 0 and bf is not None:
            x = adjust_brightness(x, bf)
        elif fn_id == 2 and sf is not None:
            x = adjust_saturation(x, sf)
    return x
