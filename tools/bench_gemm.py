import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M = 256 * 197
shapes = [("sq4k", 4096, 4096, 4096), ("sq8k", 8192, 8192, 8192), ("qkv", M, 2304, 768), ("fc1", M, 3072, 768), ("fc2/proj-in", M, 768, 3072),
          ("proj", M, 768, 768), ("lmhead", 25000, 8192, 768), ("dlm", 25000, 768, 8192)]
for name, m, n, k in shapes:
    A = (torch.randn(m, k, device="cuda")).bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o))
    line = f"{name:12s} M={m:6d} N={n:5d} K={k:5d}  bias_bf16 {2*m*n*k/dt/1e12:7.1f} TF ({dt*1e6:7.1f} us)"
    if name in ("fc1",):
        o2 = torch.empty_like(o); bias = torch.randn(n, device="cuda")
        dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=o, out1=o2, bias=bias))
        line += f" | gelu {2*m*n*k/dt/1e12:7.1f} TF"
    if name in ("proj", "fc2/proj-in"):
        x = torch.randn(m, n, device="cuda"); x2 = torch.empty_like(x); g = torch.randn(n, device="cuda"); bias = torch.randn(n, device="cuda")
        dt = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_RESIDUAL, out0=o, bias=bias, vec1=g, resid=x2, aux=x, ldaux=n, rows_per_sample=197))
        line += f" | residual {2*m*n*k/dt/1e12:7.1f} TF"
    print(line, flush=True)
for name, r, n, k in [("w_qkv", M, 2304, 768), ("w_fc1", M, 3072, 768), ("w_fc2", M, 768, 3072), ("w_proj", M, 768, 768), ("w_lm", 25000, 8192, 768)]:
    A = torch.randn(r, n, device="cuda").bfloat16(); B = torch.randn(r, k, device="cuda").bfloat16()
    o = torch.zeros(n, k, device="cuda")
    dt = t(lambda: ops.gemm_tn(A, B, r, n, k, o, accumulate=True))
    ws = torch.empty(max(ops.gemm_tn_workspace(r, n, k), 16), dtype=torch.uint8, device="cuda")
    dt2 = t(lambda: ops.gemm_tn(A, B, r, n, k, o, accumulate=True, workspace=ws))
    print(f"{name:12s} R={r:6d} N={n:5d} K={k:5d}  tn {2*r*n*k/dt/1e12:7.1f} TF ({dt*1e6:7.1f} us) | workspace {2*r*n*k/dt2/1e12:7.1f} TF ({dt2*1e6:7.1f} us)", flush=True)
