"""Does desynchronising the workgroups make the deferred stores cheap?  gemm_p8d / gemm_p8 with option gemm_stagger < 0
(every workgroup starts late by a hashed fraction of |stagger| * nk cycles): kernel time incl. the delay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
def timed(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 256 * 192
for (n, k, epi) in ((3072, 768, "bias"), (3072, 768, "gelu"), (2304, 768, "bias"), (1536, 3072, "bias")):
    A = torch.randn(M, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(M, n, dtype=torch.bfloat16, device="cuda"); o2 = torch.empty_like(o); bias = torch.randn(n, device="cuda")
    def call():
        if epi == "bias": ops.gemm_nt(A, B, M, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias)
        else: ops.gemm_nt(A, B, M, n, k, ops.EPI_BIAS_GELU, out0=o, out1=o2, bias=bias)
    for d in (1, 0):
        _lib.set_option("gemm_p8d", d)
        row = []
        for st in (0, -300, -600, -1200, -2400, -3600):
            _lib.set_option("gemm_stagger", st)
            row.append(f"{st}: {timed(call):.1f}")
        print(f"N={n} K={k} {epi} p8d={d} | " + " | ".join(row), flush=True)
_lib.set_option("gemm_stagger", 0); _lib.set_option("gemm_p8d", 1)
