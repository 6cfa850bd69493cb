"""Event record contract -- mirror of the decoders / loaders that produce the (N,4) float64 rows [x, y, t, p] every
later stage reads:

  * N-Caltech101 5-byte records      /root/reference/process_data/process_dataset.py:48-63
  * imgnet_npy_loader                /root/reference/mem/dataset_folder.py:285-292
  * dsec_npy_loader                  /root/reference/mem/dataset_folder.py:275-283
  * caltech_npy_loader / ncars_npy_loader   :294-302 (already (N,4): passed through)

MI355X design: the reference decodes one record per Python loop iteration (offline) and converts columns with NumPy
on DataLoader workers.  Here the file bytes are uploaded once and the conversion is a HIP pass (csrc/records.hip), so
the events are born in HBM in the layout the fused rasterizer consumes.  Functions return CUDA float64 tensors
[N,4]; ``.cpu().numpy()`` gives the reference's ndarray bit for bit.  No CPU fallback.
"""
import numpy as np
import torch

from ._lib import C, check, declare, f64, i32, i64, lib, ptr, require_gpu, stream_ptr, sz, vp

declare({
    "memhip_decode_ncaltech101": (i32, [vp, i64, vp, vp]),
    "memhip_events_from_columns": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, i64, vp, vp]),
    "memhip_events_dsec_workspace": (sz, [i64]),
    "memhip_events_dsec": (i32, [vp, i32, i64, f64, vp, vp, vp, sz, vp]),
})

_DT = {np.dtype(k): v for k, v in {"uint8": 0, "int8": 1, "uint16": 2, "int16": 3, "uint32": 4, "int32": 5, "uint64": 6,
                                   "int64": 7, "float32": 8, "float64": 9, "bool": 10}.items()}


def _dev_bytes(a):
    """Host ndarray -> CUDA uint8 tensor with the same bytes (torch has no uint16/32/64 arithmetic: bytes only)."""
    a = np.ascontiguousarray(a)
    if a.dtype.byteorder == ">":
        a = a.astype(a.dtype.newbyteorder("<"))
    if a.dtype not in _DT:
        raise TypeError(f"unsupported column dtype {a.dtype}")
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).cuda(), _DT[a.dtype]


def decode_ncaltech101(raw):
    """bytes / uint8 ndarray / uint8 tensor of 5-byte records -> f64 [N,4] (cuda).  A trailing partial record raises
    (the reference's ``data[2]`` on the short read raises IndexError)."""
    require_gpu()
    if isinstance(raw, (bytes, bytearray, memoryview)):
        raw = np.frombuffer(raw, dtype=np.uint8)
    if isinstance(raw, np.ndarray):
        raw = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.uint8).copy())
    raw = raw.to("cuda", dtype=torch.uint8).contiguous()
    n = raw.numel()
    if n % 5:
        raise IndexError("truncated N-Caltech101 record: %d bytes is not a multiple of 5" % n)
    ev = torch.empty((n // 5, 4), dtype=torch.float64, device="cuda")
    check(lib.memhip_decode_ncaltech101(ptr(raw), n, ptr(ev), stream_ptr()), "decode_ncaltech101")
    return ev


def ncaltech101_file(path):
    """One N-Caltech101 ``.bin`` file -> f64 [N,4] (cuda): what process_dataset.py:46-63 saves as ``.npy``."""
    with open(path, "rb") as f:
        return decode_ncaltech101(f.read())


def events_from_columns(x, y, t, p):
    """Column arrays (host ndarrays, any integer / float dtype; p bool or integer) -> f64 [N,4] = [x, y, t, 2p-1]
    with the int8 arithmetic of dataset_folder.py:288-289."""
    require_gpu()
    n = len(x)
    assert len(y) == n and len(t) == n and len(p) == n
    (xd, xc), (yd, yc), (td, tc), (pd, pc) = _dev_bytes(x), _dev_bytes(y), _dev_bytes(t), _dev_bytes(p)
    ev = torch.empty((n, 4), dtype=torch.float64, device="cuda")
    check(lib.memhip_events_from_columns(ptr(xd), xc, ptr(yd), yc, ptr(td), tc, ptr(pd), pc, n, ptr(ev), stream_ptr()),
          "events_from_columns")
    return ev


def dsec_rows_to_events(data, y_limit=440.0):
    """(N,4) ndarray [x, y, t, p in {0,1}] -> f64 [N',4] with p <- 2p-1 and the rows y >= 440 dropped, order kept."""
    require_gpu()
    data = np.asarray(data)
    assert data.ndim == 2 and data.shape[1] == 4
    n = data.shape[0]
    d, code = _dev_bytes(data)
    out = torch.empty((n, 4), dtype=torch.float64, device="cuda")
    n_out = torch.zeros(1, dtype=torch.int64, device="cuda")
    wsb = lib.memhip_events_dsec_workspace(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    check(lib.memhip_events_dsec(ptr(d), code, n, float(y_limit), ptr(out), ptr(n_out), ptr(ws), wsb, stream_ptr()),
          "events_dsec")
    return out[: int(n_out.item())]


# ---- loaders with the reference's names (mem/dataset_folder.py:275-302)
def dsec_npy_loader(path):
    return dsec_rows_to_events(np.load(path))


def imgnet_npy_loader(path):
    data = np.load(path)
    return events_from_columns(data["x"], data["y"], data["t"], data["p"])


def caltech_npy_loader(path):
    require_gpu()
    return torch.from_numpy(np.load(path)).cuda()


ncars_npy_loader = caltech_npy_loader
