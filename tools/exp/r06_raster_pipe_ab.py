"""A/B of the option `raster_pipe` (raster_bin_keys requests its next chunk before sorting the current one), interleaved on
one box: 64 x 1 M events on 480 x 640 (BASELINE configs[3]), uniform and hot-pixel streams, outputs compared bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mem_amd import datasets as D, _lib
B = int(os.environ.get("B", 64)); n = int(os.environ.get("N", 1_000_000)); H, W = 480, 640
OPTS = os.environ.get("OPTS", "raster_pipe").split(",")
VALS = [int(v) for v in os.environ.get("VALS", "0,1").split(",")]
g = torch.Generator(device="cuda").manual_seed(4)
def make(hotfrac):
    x = torch.randint(0, W, (B * n,), generator=g, device="cuda"); y = torch.randint(0, H, (B * n,), generator=g, device="cuda")
    if hotfrac:
        hot = torch.rand((B * n,), generator=g, device="cuda") < hotfrac
        hp = torch.randint(0, 16, (B * n,), generator=g, device="cuda")
        x = torch.where(hot, 100 + 7 * hp, x); y = torch.where(hot, 50 + 3 * hp, y)
    t = torch.rand((B * n,), generator=g, device="cuda", dtype=torch.float64) * 3e5
    p = torch.randint(0, 2, (B * n,), generator=g, device="cuda") * 2 - 1
    return torch.stack([x.double(), y.double(), t, p.double()], 1).contiguous()
off = torch.arange(0, B + 1, device="cuda", dtype=torch.int64) * n
def timed(f, k=20):
    f(); f(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / k * 1e-3
byts = B * (32 * n + 3 * H * W)
for name, hf in (("uniform", 0.0), ("hot-pixel 1%", 0.01)):
    ev = make(hf); ref = None
    for rep in range(3):
        for v in VALS:
            for o in OPTS: _lib.set_option(o, v)
            out = D.rasterize(ev, off, H, W, False, strict=False, binned=True)
            if ref is None: ref = out.clone()
            same = bool(torch.equal(out, ref))
            dt = timed(lambda: D.rasterize(ev, off, H, W, False, strict=False, binned=True))
            print(f"{name:13s} {'+'.join(OPTS)}={v}: {dt*1e6:7.1f} us  {byts/dt/8e12:5.3f} of 8 TB/s  equal={same}", flush=True)
