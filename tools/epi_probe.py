"""Epilogue cost probe: the MLP GEMM shapes with each fused epilogue (A/B builds via MEMHIP_LIB)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
M = 256 * 197
def run(name, m, n, k):
    A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); o2 = torch.empty_like(o)
    aux = torch.randn(m, n, device="cuda").bfloat16(); bias = torch.randn(n, device="cuda"); cs = torch.zeros(n, device="cuda")
    x = torch.randn(m, n, device="cuda"); x2 = torch.empty_like(x); g = torch.randn(n, device="cuda")
    fl = 2 * m * n * k
    r = {}
    r["plain"] = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o))
    r["bias"] = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias))
    r["gelu"] = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=o, out1=o2, bias=bias))
    r["dgelu"] = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_DGELU, out0=o, aux=aux, colsum=cs))
    r["mulaux"] = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_MUL_AUX, out0=o, aux=aux, colsum=cs))
    r["resid"] = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_RESIDUAL, out0=None, bias=bias, vec1=g, resid=x2, aux=x, ldaux=n, rows_per_sample=197))
    print(f"{name:8s} M={m} N={n} K={k} " + " ".join(f"{a} {v*1e6:6.1f}us/{fl/v/1e12:5.0f}TF" for a, v in r.items()), flush=True)
from mem_amd import _lib
for kv in os.environ.get('OPTS', '').split(','):
    if kv: _lib.set_option(kv.split('=')[0], int(kv.split('=')[1]))
for st in [int(v) for v in os.environ.get('STAGGERS', '0').split(',')]:
  _lib.set_option("gemm_stagger", st); print("gemm_stagger", st)
  for a in sys.argv[1:] or ["fc1", "fc2", "proj", "qkv"]:
      if a == "fc1": run("fc1", M, 3072, 768)
      if a == "fc2": run("fc2", M, 768, 3072)
      if a == "proj": run("proj", M, 768, 768)
      if a == "qkv": run("qkv", M, 2304, 768)
      if a == "fc1s": run("fc1s", 43520, 3072, 768)
