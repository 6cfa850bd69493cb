"""Grouped weight-gradient launch (memhip_gemm_bf16_tn_group) against the single calls: results and time, ViT-B shapes
(M = 256 x 197 rows).  usage: tn_group_check.py [time]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib

do_time = len(sys.argv) > 1
M, D, Hd = 256 * 197, 768, 3072
g = torch.Generator(device="cuda").manual_seed(3)
def rnd(r, c): return (torch.randn(r, c, generator=g, device="cuda") * 0.5).bfloat16()
groups = {
    "proj+qkv": [(rnd(M, D), rnd(M, D), M, D, D), (rnd(M, 3 * D), rnd(M, D), M, 3 * D, D)],
    "fc2+fc1": [(rnd(M, D), rnd(M, Hd), M, D, Hd), (rnd(M, Hd), rnd(M, D), M, Hd, D)],
    "ragged rows": [(rnd(M - 1000, D), rnd(M - 1000, D), M - 1000, D, D), (rnd(M - 1000, 3 * D), rnd(M - 1000, D), M - 1000, 3 * D, D)],
    "different R": [(rnd(5000, 256), rnd(5000, 512), 5000, 256, 512), (rnd(9000, 512), rnd(9000, 256), 9000, 512, 256),
                    (rnd(3000, 256), rnd(3000, 256), 3000, 256, 256)],
}
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, probs in groups.items():
    shapes = [(R, N, K) for _, _, R, N, K in probs]
    wsb = max(ops.gemm_tn_group_workspace(shapes), max(ops.gemm_tn_workspace(*sh) for sh in shapes))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    outs_1 = [torch.zeros(N, K, device="cuda") for _, _, _, N, K in probs]
    outs_g = [torch.full((N, K), 7.0, device="cuda") for _, _, _, N, K in probs]
    def single(acc=False):
        for (A, B, R, N, K), o in zip(probs, outs_1): ops.gemm_tn(A, B, R, N, K, o, accumulate=acc, workspace=ws)
    def grouped(acc=False):
        ops.gemm_tn_group([(A, B, R, N, K, o) for (A, B, R, N, K), o in zip(probs, outs_g)], accumulate=acc, workspace=ws)
    single(); grouped(); torch.cuda.synchronize()
    for (A, B, R, N, K), o1, og in zip(probs, outs_1, outs_g):
        ref = A[:R].float().t() @ B[:R].float()
        e1 = ((o1 - ref).norm() / ref.norm()).item(); eg = ((og - ref).norm() / ref.norm()).item()
        d = ((o1 - og).abs().max() / ref.abs().max()).item()
        print(f"{name} R={R} N={N} K={K}: single vs fp32 matmul {e1:.2e}  grouped {eg:.2e}  max|single-grouped|/max|ref| {d:.2e}", flush=True)
    # accumulate = 1 and run-to-run determinism
    keep = [o.clone() for o in outs_g]
    grouped(acc=True); torch.cuda.synchronize()
    acc_ok = all(torch.allclose(o, 2 * k, rtol=1e-6, atol=0) for o, k in zip(outs_g, keep))
    grouped(); torch.cuda.synchronize()
    det = all(torch.equal(o, k) for o, k in zip(outs_g, keep))
    print(f"{name}: accumulate doubles {acc_ok}  bit-identical repeat {det}", flush=True)
    if do_time:
        for rep in range(2):
            t1, tg = tm(single), tm(grouped)
            _lib.lib.memhip_set_option(b"tn_group", 0); t0 = tm(grouped); _lib.lib.memhip_set_option(b"tn_group", 1)
            print(f"{name}: single calls {t1:.1f} us  grouped {tg:.1f} us  grouped entry with tn_group=0 {t0:.1f} us", flush=True)
