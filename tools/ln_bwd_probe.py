"""layernorm_bwd_branch at ViT-B and ViT-L widths: time per launch by grid size (option ln_bwd_grid)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for R, D in ((256 * 197, 768), (16 * 1201, 1024), (64 * 1201, 1024)):
    g = torch.Generator(device="cuda").manual_seed(1)
    dy = torch.randn(R, D, generator=g, device="cuda").bfloat16(); x = torch.randn(R, D, generator=g, device="cuda")
    gamma = torch.randn(D, generator=g, device="cuda"); mean = torch.randn(R, generator=g, device="cuda"); rstd = torch.rand(R, generator=g, device="cuda") + 0.5
    dres = torch.randn(R, D, generator=g, device="cuda"); dgamma = torch.zeros(D, device="cuda"); dbeta = torch.zeros(D, device="cuda")
    gb = torch.randn(D, generator=g, device="cuda"); dyb = torch.empty(R, D, dtype=torch.bfloat16, device="cuda")
    mb = R * D * (2 + 4 + 4 + 4 + 2) / 1e6
    res = []
    for grid in (256, 384, 512, 640, 768, 1024, 1536):
        _lib.set_option("ln_bwd_grid", grid)
        dbb = torch.zeros(D, device="cuda"); us = t(lambda: ops.layernorm_bwd_branch(dy, x, gamma, mean, rstd, dres, dgamma, dbeta, R, D, None, gb, dyb, None, dbb, rows_per_sample=197))
        res.append(f"{grid}: {us:.0f} us ({mb / us:.2f} TB/s)")
    _lib.set_option("ln_bwd_grid", 768)
    print(f"R={R} D={D} ({mb:.0f} MB): " + " | ".join(res), flush=True)
