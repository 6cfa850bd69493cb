"""The K = 768 projection with the residual epilogue (x_out = x_in + gamma * (A W^T + b), fp32 residual stream) as the engine
calls it, under the dispatch options: where does its 3.5 TB/s come from?  usage: resid_gemm_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
def t(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M = 256 * 197
for name, n, k in (("proj", 768, 768), ("fc2", 768, 3072)):
    A = torch.randn(M, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    x = torch.randn(M, n, device="cuda"); x2 = torch.empty_like(x); g = torch.randn(n, device="cuda"); bias = torch.randn(n, device="cuda")
    o = torch.empty(M, n, dtype=torch.bfloat16, device="cuda")
    byts = M * k * 2 + n * k * 2 + 2 * M * n * 4
    for opts in ({}, {"gemm_p8_pair": 0}, {"gemm_p8_half": 0}, {"gemm_p8": 0}, {"gemm_p8": 0, "gemm256": 0}):
        for kk, v in opts.items(): assert _lib.lib.memhip_set_option(kk.encode(), v) == 0
        dr = t(lambda: ops.gemm_nt(A, B, M, n, k, ops.EPI_RESIDUAL, out0=None, bias=bias, vec1=g, resid=x2, aux=x, ldaux=n, rows_per_sample=197))
        db = t(lambda: ops.gemm_nt(A, B, M, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias))
        for kk in opts: assert _lib.lib.memhip_set_option(kk.encode(), 1) == 0
        print(f"{name} {str(opts):40s} residual {dr*1e6:7.1f} us = {2*M*n*k/dr/1e12:6.1f} TF, {byts/dr/1e12:5.2f} TB/s of operand + residual bytes | bias only {db*1e6:7.1f} us", flush=True)
# phase stagger of the persistent workgroups (option gemm_stagger, cycles per K-tile; < 0: every workgroup): do desynchronised
# epilogues hide under the other CUs' main loops?
for name, n, k in (("proj", 768, 768), ("fc2", 768, 3072)):
    A = torch.randn(M, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    x = torch.randn(M, n, device="cuda"); x2 = torch.empty_like(x); g = torch.randn(n, device="cuda"); bias = torch.randn(n, device="cuda")
    for st in (0, 500, 1000, 2000, 4000, -500, -1000, -2000, 0):
        assert _lib.lib.memhip_set_option(b"gemm_stagger", st) == 0
        dr = t(lambda: ops.gemm_nt(A, B, M, n, k, ops.EPI_RESIDUAL, out0=None, bias=bias, vec1=g, resid=x2, aux=x, ldaux=n, rows_per_sample=197))
        print(f"{name} gemm_stagger {st:6d}: residual {dr*1e6:7.1f} us", flush=True)
    assert _lib.lib.memhip_set_option(b"gemm_stagger", 0) == 0
