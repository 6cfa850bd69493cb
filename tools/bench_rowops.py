"""Micro-benchmark of the row kernels of the ViT-B step at B = 256 (HIP events, 20 reps): the fused LayerNorm-backward +
branch-backward pass against its workgroup count (every workgroup ends in one atomic per column and output array: atomics on
one address serialise), the column sums and the patch-embedding backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
dev = "cuda"
B, T, D, L = 256, 197, 768, 196
M = B * T
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
dy = torch.randn(M, D, device=dev).bfloat16(); x = torch.randn(M, D, device=dev); g = torch.randn(D, device=dev)
mean = x.mean(1).contiguous(); rstd = (1.0 / x.std(1)).contiguous(); dres = torch.randn(M, D, device=dev)
dg, db, dgb, dbb = (torch.zeros(D, device=dev) for _ in range(4)); gb = torch.randn(D, device=dev); dyb = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
for grid in (768, 512, 384, 256, 192, 128):
    _lib.set_option("ln_bwd_grid", grid)
    us = t(lambda: ops.layernorm_bwd_branch(dy, x, g, mean, rstd, dres, dg, db, M, D, None, gb, dyb, None, dbb, rows_per_sample=T))
    print(f"ln_bwd_branch grid {grid:4d}: {us:7.1f} us  ({(M * D * (2 + 4 + 8 + 2)) / us / 1e6:5.2f} TB/s)")
# the final-norm backward as the step runs it: 25 088 masked-token rows gathered from / scattered to the residual stream
Mm = 25088
ridx = torch.randperm(M, device=dev)[:Mm].sort().values.int()
dyn = torch.randn(Mm, D, device=dev).bfloat16(); mn = torch.randn(Mm, device=dev); rn = torch.rand(Mm, device=dev) + 0.5
for grid in (768, 576, 384, 192, 96):
    _lib.set_option("ln_bwd_grid", grid)
    us = t(lambda: ops.layernorm_bwd(dyn, x, g, mn, rn, dres, dg, db, Mm, D, accumulate=False, row_idx=ridx))
    print(f"ln_bwd (gathered rows) workgroups {grid * 4 // 3:5d}: {us:7.1f} us  ({Mm * D * (2 + 4 + 4) / us / 1e6:5.2f} TB/s)")
_lib.set_option("ln_bwd_grid", 768)
for (R, Cc, name) in ((B * L, D, "patch-embed bias"), (25088, 8192, "lm_head bias"), (M, 512, "C=512")):
    a = torch.randn(R, Cc, device=dev).bfloat16(); o = torch.zeros(Cc, device=dev)
    us = t(lambda: ops.colsum_bf16(a, R, Cc, o))
    print(f"colsum [{R}, {Cc}] ({name}): {us:7.1f} us  ({R * Cc * 2 / us / 1e6:5.2f} TB/s)")
    o.zero_(); ops.colsum_bf16(a, R, Cc, o); ref = a.float().sum(0)
    print("   max rel err", float(((o - ref).abs() / (ref.abs() + 1.0)).max()))
dx = torch.randn(M, D, device=dev); mask = (torch.rand(B * L, device=dev) < 0.5).to(torch.uint8)
dy2 = torch.empty(B * L, D, device=dev, dtype=torch.bfloat16); dcls = torch.zeros(D, device=dev); dmt = torch.zeros(D, device=dev)
us = t(lambda: ops.embed_bwd(dx, mask, B, L, D, dy2, dcls, dmt))
print(f"embed_bwd: {us:7.1f} us  ({(M * D * 4 + B * L * D * 2) / us / 1e6:5.2f} TB/s)")
