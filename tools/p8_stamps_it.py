"""Per-ITERATION timeline of gemm_p8 (diagnostic build: tools/build_variant.sh stampit -DP8_STAMP=2; MEMHIP_LIB=variants/stampit.so):
cycles between consecutive loop iterations (two K-tiles each) of a workgroup; the first iteration of a tile contains the previous
tile's epilogue."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
M = 256 * 197
def run(name, m, n, k, epi):
    A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(m, n, dtype=torch.bfloat16, device="cuda"); o2 = torch.empty_like(o)
    aux = torch.randn(m, n, device="cuda").bfloat16(); bias = torch.randn(n, device="cuda")
    def call():
        if epi == "bias": ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias)
        if epi == "gelu": ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=o, out1=o2, bias=bias)
        if epi == "dgelu": ops.gemm_nt(A, B, m, n, k, ops.EPI_DGELU, out0=o, aux=aux)
    for _ in range(3): call()
    torch.cuda.synchronize()
    call(); torch.cuda.synchronize()
    buf = np.zeros(256 * 32, dtype=np.uint64)
    assert _lib.lib.memhip_debug_p8_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.reshape(256, 32).astype(np.int64)
    d = np.diff(t, axis=1)                      # [256, 31]: iteration i+1 end - iteration i end
    med = np.median(d, axis=0).astype(int)
    per_tile = k // 128
    print(f"{name} {epi} N={n} K={k} ({per_tile} iterations per tile): median cycles per iteration (x = first iteration of a tile, holds the epilogue):")
    print("   " + " ".join(("x" if (i + 1) % per_tile == 0 else " ") + str(v) for i, v in enumerate(med)), flush=True)
for epi in ("bias", "gelu", "dgelu"):
    run("fc1", M, 3072, 768, epi)
run("fc2", M, 768, 3072, "bias")
