"""Per-tile timeline of gemm_p8d (diagnostic build: tools/build_variant.sh p8dstamp -DP8D_STAMP; MEMHIP_LIB=mem_amd/exp/p8dstamp.so).
Segments per tile (cycles, median over workgroups, tiles 1..6): N = first K-tile (deferred stores), M = second K-tile,
loop = the remaining K-tiles, realign = barrier wait before the arithmetic, math = the arithmetic, gap = math end -> next tile start."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
def run(m, n, k, epi):
    A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); o2 = torch.empty_like(o)
    aux = torch.randn(m, n, device="cuda").bfloat16(); bias = torch.randn(n, device="cuda")
    x = torch.randn(m + 512, n, device="cuda"); xo = torch.empty_like(x)
    def call():
        if epi == "bias": ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias)
        if epi == "gelu": ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=o, out1=o2, bias=bias)
        if epi == "dgelu": ops.gemm_nt(A, B, m, n, k, ops.EPI_DGELU, out0=o, aux=aux)
        if epi == "resid": ops.gemm_nt(A, B, m, n, k, ops.EPI_RESIDUAL, bias=bias, resid=xo, aux=x, ldaux=n, rows_per_sample=197)
    for _ in range(3): call()
    torch.cuda.synchronize()
    call(); torch.cuda.synchronize()
    buf = np.zeros(256 * 2 * 8 * 6, dtype=np.uint64)
    assert _lib.lib.memhip_debug_p8d_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(256, 2, 8, 6).astype(np.int64)
    names = ["N", "M", "loop", "realign", "math", "gap"]
    nt = min(7, (m // 256) * (n // 256) // 256 - 1)           # whole rounds this launch has
    for g in (0, 1):
        tt = t[:, g]                                           # [256, 8, 6]
        a, b = 1, nt
        seg = np.stack([tt[:, a:b, 1] - tt[:, a:b, 0], tt[:, a:b, 2] - tt[:, a:b, 1], tt[:, a:b, 3] - tt[:, a:b, 2],
                        tt[:, a:b, 4] - tt[:, a:b, 3], tt[:, a:b, 5] - tt[:, a:b, 4], tt[:, a + 1:b + 1, 0] - tt[:, a:b, 5]], -1)
        med = np.median(seg.reshape(-1, 6), axis=0).astype(int)
        tile = np.median((tt[:, a + 1:b + 1, 0] - tt[:, a:b, 0]).reshape(-1)).astype(int)
        print(f"{epi} N={n} K={k} waves {4*g}-{4*g+3}: " + " ".join(f"{x}={y}" for x, y in zip(names, med)) + f" | tile={tile} ({k//64} K-tiles, {nt} rounds)", flush=True)
M = 256 * 192
for epi in ("bias", "gelu", "dgelu"):
    run(M, 3072, 768, epi)
run(M, 2304, 768, "bias")
run(M, 768 * 2, 768, "bias")
run(M, 768 * 2, 768, "resid")
run(M, 768 * 2, 3072, "bias")
