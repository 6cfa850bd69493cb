"""NumPy restatement of the reference's event-stream path.  TEST INFRASTRUCTURE.

Everything here follows /root/reference/mem/datasets.py, mem/dataset_folder.py
and process_data/process_dataset.py (file:line cited per function) and is
pinned against those by oracle/gen_golden.py -> tests/golden/events_*.npz.

Events are ``(N, 4) float64`` rows ``[x, y, t, p]`` with ``p`` in {-1, +1}
(mem/dataset_folder.py:275-302).
"""
import numpy as np


# --------------------------------------------------------------------------
# rasterizer -- mem/datasets.py:552-595 (EventArrToImg.__call__)
# --------------------------------------------------------------------------
def event_arr_to_img(ev, H=None, W=None, time_surface=False):
    """(N,4) f64 -> (H, W, 3) uint8 ``[pos, tss, neg]``.

    datasets.py:567-569  x,y are truncated toward zero (astype(int)).
    datasets.py:571-575  H/W default to max+1 of the truncated coordinates.
    datasets.py:581-582  unbuffered +1 scatter into uint8 => counts mod 256;
                         only p == +1 / p == -1 events count.
    datasets.py:587-590  optional time surface: (t-tmin)/max(t-tmin)*255 cast
                         to uint8, duplicate pixels resolved by array order
                         (last event wins).
    """
    ev = np.asarray(ev, dtype=np.float64)
    xs = ev[:, 0].astype(np.int64)
    ys = ev[:, 1].astype(np.int64)
    ts = ev[:, 2]
    ps = ev[:, 3]
    if W is None:
        W = int(xs.max()) + 1
    if H is None:
        H = int(ys.max()) + 1
    flat = xs + W * ys
    # NumPy index semantics of the reference's np.add.at / fancy assignment:
    # an index in [-H*W, 0) wraps from the end, anything else outside
    # [0, H*W) raises IndexError.  (x >= W with a fixed W silently lands in
    # the next row -- also reference behaviour.)
    if flat.size and (flat.min() < -H * W or flat.max() >= H * W):
        raise IndexError("event outside the H x W canvas")
    flat = np.where(flat < 0, flat + H * W, flat)
    pos = np.bincount(flat[ps == 1], minlength=H * W)
    neg = np.bincount(flat[ps == -1], minlength=H * W)
    img = np.zeros((3, H * W), dtype=np.uint8)
    img[0] = (pos & 0xFF).astype(np.uint8)
    img[2] = (neg & 0xFF).astype(np.uint8)
    if time_surface:
        tn = ts - ts.min()
        val = (tn / tn.max() * 255)
        # fancy-index assignment with duplicates: the last occurrence in array
        # order wins (element-wise sequential copy).
        last = np.full(H * W, -1, dtype=np.int64)
        np.maximum.at(last, flat, np.arange(flat.shape[0]))
        hit = last >= 0
        img[1, hit] = val[last[hit]].astype(np.uint8)
    return img.reshape(3, H, W).transpose(1, 2, 0)


def to_tensor_chw(img_hwc_u8):
    """torchvision ToTensor on a uint8 HWC ndarray (datasets.py:637): CHW
    float32 in [0,1] by division by 255 (torchvision arithmetic: un-vendored,
    semantics = ``img.permute(2,0,1).float().div(255)``)."""
    return (np.ascontiguousarray(img_hwc_u8.transpose(2, 0, 1)).astype(np.float32)
            / np.float32(255))


# --------------------------------------------------------------------------
# event-level augmentations -- mem/datasets.py:464-549, 598-609.
# The random draws are arguments so that parity can be checked draw-for-draw.
# --------------------------------------------------------------------------
def reshape_scale_xy(ev, newH=224, newW=224, oldH=480, oldW=640, is_train=False):
    """datasets.py:464-485: scale x,y (float multiply; truncation happens later
    in the rasterizer).  Train: both by 256/min(oldH,oldW); eval: new/old."""
    ev = np.array(ev, dtype=np.float64, copy=True)
    if is_train:
        s = 256 / [oldH, oldW][int(np.argmin([oldH, oldW]))]
        sx = sy = s
    else:
        sx, sy = newW / oldW, newH / oldH
    ev[:, 0] *= sx
    ev[:, 1] *= sy
    return ev


def slice_random_max_evs(ev, keep_max, rand_start=None):
    """datasets.py:488-498: contiguous window of at most ``keep_max`` events,
    start = random.choice(range(len-keep+1)) (here: ``rand_start``)."""
    if len(ev) > keep_max:
        ev = ev[rand_start:rand_start + keep_max, :]
    return ev


def random_time_flip(ev, u, p=0.5):
    """datasets.py:598-609: if u < p: reverse order, t <- t_first_after_flip - t,
    p <- -p."""
    if u < p:
        ev = np.flip(ev, axis=0).copy()
        ev[:, 2] = ev[0, 2] - ev[:, 2]
        ev[:, 3] = -ev[:, 3]
    return ev


def flip_along_x(ev, u, W=None, p=0.5):
    """datasets.py:501-521: W defaults to trunc(max x)+1; if u < p: x <- W-1-x."""
    ev = np.array(ev, dtype=np.float64, copy=True)
    if W is None:
        W = ev[:, 0].max().astype(np.int64) + 1
    if u < p:
        ev[:, 0] = W - 1 - ev[:, 0]
    return ev


def random_shift(ev, x_shift, y_shift, H=None, W=None):
    """datasets.py:524-549: bounds from max()+1 BEFORE the shift when not
    fixed; integer shift; keep events with 0<=x<W and 0<=y<H."""
    ev = np.array(ev, dtype=np.float64, copy=True)
    if W is None:
        W = ev[:, 0].max().astype(np.int64) + 1
    if H is None:
        H = ev[:, 1].max().astype(np.int64) + 1
    ev[:, 0] += x_shift
    ev[:, 1] += y_shift
    ok = (ev[:, 0] >= 0) & (ev[:, 0] < W) & (ev[:, 1] >= 0) & (ev[:, 1] < H)
    return ev[ok]


# --------------------------------------------------------------------------
# raw record contract -- process_data/process_dataset.py:48-63 (N-Caltech101)
# and mem/dataset_folder.py:275-292 (loaders)
# --------------------------------------------------------------------------
def decode_ncaltech101(raw):
    """5-byte records -> (N,4) f64.  process_dataset.py:52-60: byte0 -> col 0,
    byte1 -> col 1, polarity = bit 7 of byte2 mapped to 2p-1, timestamp = low 7
    bits of byte2 : byte3 : byte4 (23 bits big-endian).  A trailing partial
    record is read by the reference's ``file.read(5)`` and would raise; here it
    is rejected the same way."""
    b = np.frombuffer(raw, dtype=np.uint8)
    if b.size % 5:
        raise IndexError("truncated N-Caltech101 record")
    b = b.reshape(-1, 5).astype(np.int64)
    pol = (b[:, 2] >> 7) & 1
    t = ((b[:, 2] & 0x7F) << 16) | (b[:, 3] << 8) | b[:, 4]
    out = np.empty((b.shape[0], 4), dtype=np.float64)
    out[:, 0] = b[:, 0]
    out[:, 1] = b[:, 1]
    out[:, 2] = t
    out[:, 3] = 2.0 * pol - 1.0
    return out


def imgnet_struct_to_events(x, y, t, p):
    """dataset_folder.py:285-292: structured N-ImageNet arrays -> [x,y,t,2p-1]."""
    ps = np.asarray(p).astype(np.int8) * 2 - 1
    return np.vstack([x, y, t, ps]).T.astype(float)


def dsec_to_events(data):
    """dataset_folder.py:275-283: p <- 2p-1, drop rows with y >= 440."""
    data = np.asarray(data).astype(float)
    data[:, 3] = 2 * data[:, 3] - 1
    return data[data[:, 1] < 440]
