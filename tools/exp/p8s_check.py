"""gemm_p8s (256x128 tiles, streamed epilogue) against gemm_p8 on the ViT-B shapes: results (bit-equal expected: same K
order, same epilogue arithmetic) and time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
M = 256 * 197
shapes = [(M, 3072, 768), (M, 2304, 768), (M, 768, 768), (M, 768, 3072), (M, 768, 2304), (M, 768, 256), (4096, 256, 512), (25088, 8192, 768)]
if len(sys.argv) > 1 and sys.argv[1] == 'mae':
    shapes = [(M, 2048, 512), (M, 3072, 512), (M, 512, 512), (M, 1024, 512), (M, 512, 1024), (M, 512, 2048), (M, 768, 512), (12800, 3072, 768), (12800, 768, 3072)]
elif len(sys.argv) > 1: shapes = shapes[: int(sys.argv[1])]
for (m, n, k) in shapes:
    torch.manual_seed(m + n + k)
    A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(n, device="cuda")
    res = {}
    for mode in (0, 1):
        _lib.set_option("gemm_p8s", mode)
        o = torch.full((m, n), 3.0, dtype=torch.bfloat16, device="cuda"); o2 = torch.full((m, n), 3.0, dtype=torch.bfloat16, device="cuda")
        g1 = torch.full((m, n), 3.0, dtype=torch.bfloat16, device="cuda"); g2 = torch.full((m, n), 3.0, dtype=torch.bfloat16, device="cuda")
        ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias)
        ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o2)
        ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=g1, out1=g2, bias=bias)
        torch.cuda.synchronize()
        tb = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias))
        tg = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=g1, out1=g2, bias=bias))
        res[mode] = (o.clone(), o2.clone(), g1.clone(), g2.clone(), tb, tg)
    _lib.set_option("gemm_p8s", 0)
    eq = [bool(torch.equal(res[0][i], res[1][i])) for i in range(4)]
    md = [float((res[0][i].float() - res[1][i].float()).abs().max()) for i in range(4)]
    fl = 2.0 * m * n * k
    print(f"M={m} N={n} K={k}: equal {eq} maxdiff {md} | bias p8 {res[0][4]:.1f} us ({fl/res[0][4]/1e6:.0f} TF) p8s {res[1][4]:.1f} us ({fl/res[1][4]/1e6:.0f} TF) | "
          f"gelu p8 {res[0][5]:.1f} us p8s {res[1][5]:.1f} us", flush=True)
