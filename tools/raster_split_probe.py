"""Does running pass 2 of one half of the batch under pass 1 of the other half pay?  The binned rasterizer on 32 / 64 x 1 M
events: one call vs the batch in G groups alternating between two HIP streams (each group: keys -> accumulate, in order)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import datasets as D
H, W, n = 480, 640, 1_000_000
g = torch.Generator(device="cuda").manual_seed(4)
def make(B):
    x = torch.randint(0, W, (B * n,), generator=g, device="cuda"); y = torch.randint(0, H, (B * n,), generator=g, device="cuda")
    t = torch.rand((B * n,), generator=g, device="cuda", dtype=torch.float64) * 3e5
    p = torch.randint(0, 2, (B * n,), generator=g, device="cuda") * 2 - 1
    return torch.stack([x.double(), y.double(), t, p.double()], 1).contiguous()
side = torch.cuda.Stream()
def split(ev, off, B, G):
    outs = []
    main = torch.cuda.current_stream()
    e0 = torch.cuda.Event(); e0.record()
    side.wait_event(e0)
    per = B // G
    for gi in range(G):
        o = off[gi * per:(gi + 1) * per + 1]
        st = main if gi % 2 == 0 else side
        with torch.cuda.stream(st):
            outs.append(D.rasterize(ev, o, H, W, False, strict=False, binned=True))
    e1 = torch.cuda.Event(); e1.record(side)
    main.wait_event(e1)
    return outs
def t(f, k=10):
    f(); f(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / k * 1e-3
for B in (32, 64):
    ev = make(B); off = torch.arange(0, B + 1, device="cuda", dtype=torch.int64) * n
    byts = B * (32 * n + 3 * H * W)
    ref = D.rasterize(ev, off, H, W, False, strict=False, binned=True)
    dt = t(lambda: D.rasterize(ev, off, H, W, False, strict=False, binned=True))
    print(f"B={B} one call: {dt*1e3:.3f} ms = {byts/dt/8e12:.3f} of 8 TB/s", flush=True)
    for G in (2, 4, 8):
        got = torch.cat(split(ev, off, B, G)); torch.cuda.synchronize()
        assert torch.equal(got, ref)
        dt = t(lambda: split(ev, off, B, G))
        print(f"B={B} {G} groups on two streams: {dt*1e3:.3f} ms = {byts/dt/8e12:.3f} of 8 TB/s", flush=True)
    del ev
