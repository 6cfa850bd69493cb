import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mem_amd import datasets as D, _lib
B, n, H, W = int(os.environ.get("B", 64)), 1_000_000, 480, 640
g = torch.Generator(device="cuda").manual_seed(4)
x = torch.randint(0, W, (B * n,), generator=g, device="cuda"); y = torch.randint(0, H, (B * n,), generator=g, device="cuda")
t = torch.rand((B * n,), generator=g, device="cuda", dtype=torch.float64) * 3e5
p = torch.randint(0, 2, (B * n,), generator=g, device="cuda") * 2 - 1
ev = torch.stack([x.double(), y.double(), t, p.double()], 1).contiguous()
off = torch.arange(0, B + 1, device="cuda", dtype=torch.int64) * n
_lib.set_option("raster_bands", int(os.environ.get("RP", "0")))
for _ in range(12): D.rasterize(ev, off, H, W, False, strict=False, binned=True)
torch.cuda.synchronize()
