import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
m, n, k = 50432, 2304, 768
A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
o = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o)); print("normal      ", 2*m*n*k/dt/1e12)
dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, lda=0)); print("A stride 0  ", 2*m*n*k/dt/1e12)
dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, lda=0, ldb=0)); print("A,B stride 0", 2*m*n*k/dt/1e12)
dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, lda=0, ldb=0, ldo0=0)); print("A,B,out stride 0", 2*m*n*k/dt/1e12)
k = 6144
A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o)); print("K=6144 normal      ", 2*m*n*k/dt/1e12)
dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, lda=0, ldb=0, ldo0=0)); print("K=6144 all stride 0", 2*m*n*k/dt/1e12)
Z = torch.zeros(m, k, device="cuda").bfloat16(); ZB = torch.zeros(n, k, device="cuda").bfloat16()
dt = t(lambda: ops.gemm_nt(Z, ZB, m, n, k, ops.EPI_BIAS_BF16, out0=o)); print("K=6144 zeros       ", 2*m*n*k/dt/1e12)
