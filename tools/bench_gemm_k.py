import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M = 256 * 197
for N in (3072, 2304):
    for K in (128, 256, 768, 1536, 3072, 6144):
        A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        dt = t(lambda: ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_BF16, out0=o))
        print(f"N={N} K={K:5d}: {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:7.1f} TF", flush=True)
