"""BASELINE configs[3]: event -> voxel rasterizer at N-ImageNet scale (~1 M events per sample, 480 x 640),
HBM roofline = (32 B per event + 3*H*W output bytes) / time against 8 TB/s."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import datasets as D
B = int(os.environ.get("B", 64)); n = int(os.environ.get("N", 1_000_000)); H, W = 480, 640
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
def t(f, k=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
for name, hf in (("uniform", 0.0), ("hot-pixel 1%", 0.01)):
    ev = make(hf)
    for label, kw in (("binned", dict(binned=True)), ("global atomics", dict(binned=False))):
        dt = t(lambda: D.rasterize(ev, off, H, W, False, strict=False, **kw))
        byts = B * (32 * n + 3 * H * W)
        print(f"{name:14s} {label:15s} B={B} N={n}: {dt*1e3:8.3f} ms  {dt/B*1e6:7.2f} us/sample  {B*n/dt/1e9:6.2f} G events/s  "
              f"{byts/dt/1e12:5.2f} TB/s algorithmic = {byts/dt/8e12:5.3f} of 8 TB/s", flush=True)
