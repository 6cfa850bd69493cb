"""gemm_q4 (tools/exp/gemm_q4.hip, four waves) against torch and against the shipped gemm_p8: results and time per launch."""
import ctypes as C, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mem_amd import ops
q4 = C.CDLL(os.path.join(ROOT, "mem_amd", "exp", "gemm_q4.so"))
q4.q4_gemm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
bad = 0
for m, n, k in ((512, 512, 256), (2048, 1024, 768), (8192, 8192, 8192), (256 * 197, 2304, 768), (256 * 197, 3072, 768), (256 * 197, 768, 3072), (256 * 197, 768, 768)):
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    A = torch.randn(m, k, generator=g, device="cuda").bfloat16(); B = (torch.randn(n, k, generator=g, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(n, generator=g, device="cuda")
    o = torch.zeros(m, n, dtype=torch.bfloat16, device="cuda"); o2 = torch.zeros_like(o)
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: q4.q4_gemm(A.data_ptr(), k, B.data_ptr(), k, o.data_ptr(), n, bias.data_ptr(), m, n, k, st)
    rc = run(); torch.cuda.synchronize()
    assert rc == 0, rc
    ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o2, bias=bias)
    torch.cuda.synchronize()
    eq = torch.equal(o, o2)
    rel = ((o.float() - o2.float()).norm() / o2.float().norm()).item()
    bad += rel > 1e-3
    tq = t(run); tp = t(lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o2, bias=bias))
    print(f"M={m} N={n} K={k}: bit-equal to gemm_p8 {eq} (rel {rel:.1e})  q4 {tq:.1f} us = {2*m*n*k/tq/1e6:.0f} TF | p8 {tp:.1f} us = {2*m*n*k/tp/1e6:.0f} TF", flush=True)
print("MISMATCHES:", bad)
