"""Reference point only (not used by the product): time torch.matmul (hipBLASLt/rocBLAS) on the
ViT-B GEMM shapes of the pretraining step, next to memhip's kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

M = 256 * 197
shapes = [("qkv", M, 2304, 768), ("proj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072),
          ("head", 256 * 75, 8192, 768)]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(a, w.t(), out=out))
    us2 = t(lambda: ops.gemm_nt(a, w, m, n, k, ops.EPI_BIAS_BF16, out0=out))
    fl = 2.0 * m * n * k
    print(f"NT {name:5s} M={m} N={n} K={k}: blaslt {us:7.1f} us {fl/us/1e6:7.1f} TF | memhip {us2:7.1f} us {fl/us2/1e6:7.1f} TF")
    # wgrad: out[n,k] = dY[m,n]^T x[m,k]
    dy = torch.randn(m, n, device="cuda", dtype=torch.bfloat16)
    g = torch.empty(n, k, device="cuda", dtype=torch.float32)
    gb = torch.empty(n, k, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: torch.matmul(dy.t(), a, out=gb))
    us2 = t(lambda: ops.gemm_tn(dy, a, m, n, k, g, accumulate=True))
    print(f"TN {name:5s}: blaslt {us:7.1f} us {fl/us/1e6:7.1f} TF | memhip {us2:7.1f} us {fl/us2/1e6:7.1f} TF")
