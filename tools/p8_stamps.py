"""Per-tile timeline of gemm_p8 (diagnostic build: tools/build_variant.sh stamp -DP8_STAMP; MEMHIP_LIB=variants/stamp.so):
main-loop and epilogue cycles of each workgroup's first tiles, per epilogue kind."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
M = 256 * 197
def run(name, m, n, k, epi):
    A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); o2 = torch.empty_like(o)
    aux = torch.randn(m, n, device="cuda").bfloat16(); bias = torch.randn(n, device="cuda"); cs = torch.zeros(n, device="cuda")
    x = torch.randn(m, n, device="cuda"); x2 = torch.empty_like(x); g = torch.randn(n, device="cuda")
    def call():
        if epi == "bias": ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias)
        if epi == "gelu": ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=o, out1=o2, bias=bias)
        if epi == "dgelu": ops.gemm_nt(A, B, m, n, k, ops.EPI_DGELU, out0=o, aux=aux, colsum=cs)
        if epi == "resid": ops.gemm_nt(A, B, m, n, k, ops.EPI_RESIDUAL, out0=None, bias=bias, vec1=g, resid=x2, aux=x, ldaux=n, rows_per_sample=197)
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(256 * 32, dtype=np.uint64)
    assert _lib.lib.memhip_debug_p8_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(256, 32)[:, :30].reshape(256, 10, 3).astype(np.int64)
    ntile = 6 if n * (m // 256) // 256 // 256 >= 6 else 2
    main = (t[:, :ntile, 1] - t[:, :ntile, 0]); epi_c = (t[:, :ntile, 2] - t[:, :ntile, 1])
    gap = t[:, 1:ntile, 0] - t[:, :ntile - 1, 2]
    start = t[:, :ntile, 1] - t[:, :ntile, 1].min(axis=0, keepdims=True)
    print(f"{name:5s} {epi:6s} N={n} K={k}: kernel {e0.elapsed_time(e1)*1e3:6.1f} us | per tile (cycles, median over workgroups): main "
          f"{np.median(main):8.0f}  epilogue {np.median(epi_c):8.0f}  gap {np.median(gap):6.0f} | epilogue-start spread over workgroups (p5..p95) "
          f"{np.percentile(start, 5):.0f}..{np.percentile(start, 95):.0f} | main by tile {np.median(main, axis=0).astype(int).tolist()} "
          f"epi by tile {np.median(epi_c, axis=0).astype(int).tolist()}", flush=True)
for name, m, n, k in (("fc1", M, 3072, 768), ("qkv", M, 2304, 768)):
    for epi in ("bias", "gelu", "dgelu", "resid"):
        run(name, m, n, k, epi)
run("fc2", M, 768, 3072, "resid"); run("fc2", M, 768, 3072, "bias"); run("proj", M, 768, 768, "resid"); run("proj", M, 768, 768, "bias")
