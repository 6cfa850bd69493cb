"""In-kernel shader clock of gemm_p8 / gemm_tn_p8 (MI355X_MICROARCH.md, DVFS give-back item 6): a -DP8_STAMP build
(tools/build_variant.sh stamp -DP8_STAMP; MEMHIP_LIB=variants/stamp.so) stamps s_memtime (shader cycles) and s_memrealtime
(constant 100 MHz) around the main loops; after >= 2 s of back-to-back launches on random data the quotient
d(s_memtime) / d(s_memrealtime) x 100 MHz is the clock the chip holds under that load.  Writes gpurun_out/r04_clock.json."""
import ctypes as C, json, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
M = 256 * 192
out = {"method": "median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz around the main loop of one launch, "
                 "taken after >= 2 s of back-to-back launches of the same kernel on random bf16 data (stamp build)",
       "spec_clock_ghz": 2.4, "spec_peak_tflops": 2500.0, "kernels": {}}
def heat(f, secs=2.2):
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): f()
        torch.cuda.synchronize(); n += 20
    return n
for name, (n, k) in {"gemm_p8 N=2304 K=768 (qkv)": (2304, 768), "gemm_p8 N=768 K=3072 (fc2 dgrad)": (768 * 2, 3072)}.items():
    A = torch.randn(M, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    o = torch.empty(M, n, dtype=torch.bfloat16, device="cuda"); bias = torch.randn(n, device="cuda")
    f = lambda: ops.gemm_nt(A, B, M, n, k, ops.EPI_BIAS_BF16, out0=o, bias=bias)
    launches = heat(f)
    f(); torch.cuda.synchronize()
    a = np.zeros(256 * 32, dtype=np.uint64); b = np.zeros(256 * 32, dtype=np.uint64)
    assert _lib.lib.memhip_debug_p8_stamps(a.ctypes.data_as(C.c_void_p)) == 0
    assert _lib.lib.memhip_debug_p8_stamps_rt(b.ctypes.data_as(C.c_void_p)) == 0
    t = a.reshape(256, 32)[:, :30].reshape(256, 10, 3).astype(np.int64); r = b.reshape(256, 32)[:, :30].reshape(256, 10, 3).astype(np.int64)
    nt = 2
    dt = t[:, 1:nt + 1, 1] - t[:, 1:nt + 1, 0]; dr = r[:, 1:nt + 1, 1] - r[:, 1:nt + 1, 0]      # main loops of tiles 1..nt
    ghz = np.median(dt / np.maximum(dr, 1)) * 0.1
    out["kernels"][name] = {"clock_ghz": round(float(ghz), 3), "main_loop_cycles_per_tile": int(np.median(dt)),
                            "k_tiles_per_tile": k // 64, "launches_before_the_stamp": launches,
                            "at_clock_peak_tflops": round(2500.0 * float(ghz) / 2.4, 1)}
    print(name, out["kernels"][name], flush=True)
R = M
for name, (n, k) in {"gemm_tn_p8 768x3072 (fc2 weight gradient)": (768, 3072), "gemm_tn_p8 2304x768 (qkv weight gradient)": (2304, 768)}.items():
    A = torch.randn(R, n, device="cuda").bfloat16(); B = torch.randn(R, k, device="cuda").bfloat16()
    G = torch.zeros(n, k, device="cuda")
    ws = torch.empty(ops.gemm_tn_workspace(R, n, k), dtype=torch.uint8, device="cuda")
    f = lambda: ops.gemm_tn(A, B, R, n, k, G, accumulate=False, workspace=ws)
    launches = heat(f)
    f(); torch.cuda.synchronize()
    a = np.zeros(256 * 4, dtype=np.uint64)
    assert _lib.lib.memhip_debug_tnp8_stamps(a.ctypes.data_as(C.c_void_p)) == 0
    t = a.reshape(256, 4).astype(np.int64)
    ok = (t[:, 3] > t[:, 1])
    ghz = np.median((t[ok, 2] - t[ok, 0]) / (t[ok, 3] - t[ok, 1])) * 0.1
    out["kernels"][name] = {"clock_ghz": round(float(ghz), 3), "main_loop_cycles": int(np.median(t[ok, 2] - t[ok, 0])),
                            "launches_before_the_stamp": launches, "at_clock_peak_tflops": round(2500.0 * float(ghz) / 2.4, 1)}
    print(name, out["kernels"][name], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(os.environ.get("MEMHIP_CLOCK_OUT", "gpurun_out/r05_clock.json"), "w"), indent=1)
