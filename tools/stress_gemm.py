"""Race screen for the phase-interleaved GEMMs (gemm_p8 / gemm_tn_p8): exact-integer products, many
repetitions and shapes, bitwise comparison with torch.  Any LDS-DMA / ds_read ordering mistake shows up
as a wrong tile in some run.  python tools/stress_gemm.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bad = 0
g = torch.Generator(device="cuda").manual_seed(0)
nt_shapes = [(50432, 2304, 768), (50432, 768, 3072), (25000, 8192, 768), (8192, 1024, 256), (4096 + 77, 768, 128),
             (20000, 3072, 768), (50432, 768, 768)]
for (M, N, K) in nt_shapes:
    A = torch.randint(-3, 4, (M, K), generator=g, device="cuda").float()
    B = torch.randint(-3, 4, (N, K), generator=g, device="cuda").float()
    ref = A @ B.t()
    Ab, Bb = A.bfloat16(), B.bfloat16()
    C = torch.empty((M, N), device="cuda")
    for r in range(reps):
        C.fill_(-1.0)
        ops.gemm_nt(Ab, Bb, M, N, K, ops.EPI_F32, out0=C)
        if not torch.equal(C, ref):
            bad += 1
            print("NT mismatch", (M, N, K), "rep", r, "wrong elements", int((C != ref).sum()))
    print("NT", (M, N, K), "ok" if bad == 0 else "BAD", flush=True)
tn_shapes = [(50432, 2304, 768), (50432, 768, 3072), (25000, 8192, 768), (50432, 768, 768), (4099, 3072, 768), (2048 + 65, 256, 256)]
for (R, N, K) in tn_shapes:
    A = torch.randint(-2, 3, (R, N), generator=g, device="cuda").float()
    B = torch.randint(-2, 3, (R, K), generator=g, device="cuda").float()
    ref = A.t() @ B
    Ab, Bb = A.bfloat16(), B.bfloat16()
    ws = torch.empty(max(ops.gemm_tn_workspace(R, N, K), 16), dtype=torch.uint8, device="cuda")
    out = torch.empty((N, K), device="cuda")
    for r in range(reps):
        for w in (None, ws):
            out.fill_(-7.0)
            ops.gemm_tn(Ab, Bb, R, N, K, out, accumulate=False, workspace=w)
            if not torch.equal(out, ref):
                bad += 1
                print("TN mismatch", (R, N, K), "rep", r, "ws" if w is not None else "atomics", int((out != ref).sum()))
    print("TN", (R, N, K), "ok" if bad == 0 else "BAD", flush=True)
print("STRESS", "PASS" if bad == 0 else f"FAIL ({bad})")
