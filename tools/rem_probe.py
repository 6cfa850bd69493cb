"""Remainder handling of the N = 768 products: default (128-row gemm_p8 for the rows of the poorly filled last round) vs the
128x128 kernel for them (gemm_p8_half = 0) vs no split (gemm_split = 0)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
ops.set_option = _lib.set_option
def t(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for M in (197 * 256, 187 * 256, 192 * 256):
    for N, K in ((768, 768), (768, 2304), (768, 3072)):
        A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        x = torch.randn(M, N, device="cuda"); x2 = torch.empty_like(x); g = torch.randn(N, device="cuda"); bias = torch.randn(N, device="cuda")
        res = []
        for name, opts in (("default", {}), ("nt128", {"gemm_p8_half": 0}), ("nosplit", {"gemm_split": 0})):
            for k, v in opts.items(): ops.set_option(k, v)
            a = t(lambda: ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_BF16, out0=o, bias=bias))
            b = t(lambda: ops.gemm_nt(A, B, M, N, K, ops.EPI_RESIDUAL, bias=bias, vec1=g, resid=x2, aux=x, ldaux=N, rows_per_sample=197))
            for k in opts: ops.set_option(k, 1)
            res.append(f"{name} bias {a:6.1f} resid {b:6.1f}")
        print(f"M={M} N={N} K={K}: " + " | ".join(res), flush=True)
