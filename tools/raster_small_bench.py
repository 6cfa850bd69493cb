"""BASELINE configs[1] rasterizer size (256 samples x 30 000 events, 224 x 224): single-pass LDS kernel vs the two-pass binned kernels."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import datasets as D
B, n, H, W = 256, 30000, 224, 224
g = torch.Generator(device="cuda").manual_seed(1)
ev = torch.stack([torch.randint(0, W, (B * n,), generator=g, device="cuda").double(), torch.randint(0, H, (B * n,), generator=g, device="cuda").double(),
                  torch.rand((B * n,), generator=g, device="cuda", dtype=torch.float64) * 3e5, (torch.randint(0, 2, (B * n,), generator=g, device="cuda") * 2 - 1).double()], 1).contiguous()
off = torch.arange(0, B + 1, device="cuda", dtype=torch.int64) * n
def t(f, k=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
a = D.rasterize(ev, off, H, W, False, strict=False, binned=False); b = D.rasterize(ev, off, H, W, False, strict=False, binned=True)
assert torch.equal(a, b)
byts = B * (32 * n + 3 * H * W)
for label, kw in (("lds single pass", dict(binned=False)), ("binned two pass", dict(binned=True))):
    dt = t(lambda: D.rasterize(ev, off, H, W, False, strict=False, **kw))
    print(f"{label:16s} {dt*1e6:8.1f} us  {byts/dt/1e12:5.2f} TB/s algorithmic = {byts/dt/8e12:5.3f} of 8 TB/s")
