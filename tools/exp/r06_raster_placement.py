"""Does the run-to-run spread of the long-stream rasterizer (417 / 430 / 440 us between PROCESSES on one box) come from where the buffers
lie?  One process, one events tensor; the key workspace and the output are placed at a sweep of byte offsets inside one big allocation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mem_amd._lib import lib, ptr, stream_ptr
from mem_amd import datasets as D   # declares the entry points
import ctypes as C
B, n, H, W = 64, 1_000_000, 480, 640
g = torch.Generator(device="cuda").manual_seed(4)
x = torch.randint(0, W, (B * n,), generator=g, device="cuda"); y = torch.randint(0, H, (B * n,), generator=g, device="cuda")
t = torch.rand((B * n,), generator=g, device="cuda", dtype=torch.float64) * 3e5
p = torch.randint(0, 2, (B * n,), generator=g, device="cuda") * 2 - 1
big = torch.empty((3 << 30,), dtype=torch.uint8, device="cuda")            # events live INSIDE it too, at a chosen offset
off = torch.arange(0, B + 1, device="cuda", dtype=torch.int64) * n
wsb = lib.memhip_rasterize_binned_workspace(B, H, W, B * n)
status = torch.empty((B,), dtype=torch.int32, device="cuda")
base = big.data_ptr()
def run(ev_off, ws_off, out_off, k=20):
    ev = torch.frombuffer  # (unused)
    evt = big[ev_off: ev_off + B * n * 32].view(torch.float64).view(B * n, 4)
    evt.copy_(torch.stack([x.double(), y.double(), t, p.double()], 1))
    ws = C.c_void_p(base + ws_off); out = C.c_void_p(base + out_off)
    f = lambda: lib.memhip_rasterize_binned_f64(ptr(evt), ptr(off), None, B, H, W, B * n, out, ptr(status), ws, wsb, stream_ptr())
    f(); f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(k): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
GB = 1 << 30
print("events at 0, output at 2.5 GB, workspace at 2 GB + d:")
for d in (0, 4096, 65536, 1 << 20, 3 << 20, 16 << 20, 100 << 20):
    print(f"  d = {d:>10d}: {run(0, 2 * GB + d, 2 * GB + 600 * (1 << 20)):.1f} us", flush=True)
print("workspace at 2 GB, events at d:")
for d in (0, 4096, 65536, 1 << 20, 3 << 20, 16 << 20):
    print(f"  d = {d:>10d}: {run(d, 2 * GB + 64 * (1<<20), 2 * GB + 700 * (1 << 20)):.1f} us", flush=True)
