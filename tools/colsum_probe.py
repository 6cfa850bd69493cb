"""Does the fused column sum (one atomic per column, wave and tile) cost the GEMM anything?  dGELU / bias epilogues with and
without colsum on the MLP shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 256 * 197
for (n, k) in ((3072, 768), (768, 3072), (768, 768), (2304, 768)):
    A = torch.randn(M, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(M, n, dtype=torch.bfloat16, device="cuda"); aux = torch.randn(M, n, device="cuda").bfloat16()
    cs = torch.zeros(n, device="cuda"); bias = torch.randn(n, device="cuda")
    r = {}
    for rep in range(2):
        r["dgelu+colsum"] = t(lambda: ops.gemm_nt(A, B, M, n, k, ops.EPI_DGELU, out0=o, aux=aux, colsum=cs))
        ws = torch.zeros(8, n, device="cuda")
        r["dgelu+8copies+fold"] = t(lambda: (ops.gemm_nt(A, B, M, n, k, ops.EPI_DGELU, out0=o, aux=aux, colsum=ws, colsum_copies=8), ops.colsum_fold(ws, 8, n, cs)))
        r["bias+8copies+fold"] = t(lambda: (ops.gemm_nt(A, B, M, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias, colsum=ws, colsum_copies=8), ops.colsum_fold(ws, 8, n, cs)))
        ws32 = torch.zeros(32, n, device="cuda")
        r["dgelu+32copies+fold"] = t(lambda: (ops.gemm_nt(A, B, M, n, k, ops.EPI_DGELU, out0=o, aux=aux, colsum=ws32, colsum_copies=32), ops.colsum_fold(ws32, 32, n, cs)))
        r["dgelu"] = t(lambda: ops.gemm_nt(A, B, M, n, k, ops.EPI_DGELU, out0=o, aux=aux))
        r["bias+colsum"] = t(lambda: ops.gemm_nt(A, B, M, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias, colsum=cs))
        r["bias"] = t(lambda: ops.gemm_nt(A, B, M, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias))
        print(f"N={n} K={k} " + " ".join(f"{a} {v:6.1f}" for a, v in r.items()), flush=True)
