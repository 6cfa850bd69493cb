"""gemm_p8 with the operand lines touched in L2 `D` K-tiles ahead (option gemm_prefetch = D; 0 = no touches): time per
launch on the ViT-B shapes, results against gemm_p8d (independent kernel, same arithmetic: bit-equal)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
def timed(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 256 * 197
Ds = [0, 1, 2, 3, 4, 6, 8]
for (n, k, epi) in ((2304, 768, "bias"), (3072, 768, "bias"), (3072, 768, "gelu"), (768, 768, "bias"), (768, 3072, "bias"), (768, 2304, "bias"), (8192, 768, "bias")):
    m = M if n != 8192 else 25088
    torch.manual_seed(n + k)
    A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(n, device="cuda")
    o = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); o2 = torch.empty_like(o)
    def call():
        if epi == "bias": ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias)
        else: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=o, out1=o2, bias=bias)
    _lib.set_option("gemm_p8d", 1); _lib.set_option("gemm_prefetch", 0)
    call(); torch.cuda.synchronize(); ref = o.clone()
    _lib.set_option("gemm_p8d", 0)
    row = []
    for D in Ds:
        _lib.set_option("gemm_prefetch", D)
        o.zero_(); call(); torch.cuda.synchronize()
        eq = bool(torch.equal(o, ref))
        row.append(f"D={D}: {timed(call):.1f}{'' if eq else ' MISMATCH'}")
    print(f"M={m} N={n} K={k} {epi}: " + " | ".join(row), flush=True)
_lib.set_option("gemm_prefetch", 0)
