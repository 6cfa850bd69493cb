"""gemm_p8d (256x256 tiles, stores deferred into the next tile's first K-tile) against gemm_p8 on the ViT-B shapes:
results (bit-equal expected: same K order, same epilogue arithmetic) and time, every epilogue the deferred kernel takes.
usage: p8d_check.py [quick]"""
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


T = 197
M = 256 * T
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
# (M, N, K, epilogues)
cases = [(M, 2304, 768, ["bias", "bias_scale"]), (M, 3072, 768, ["bias", "gelu", "gelu_dg", "dgelu", "mul_aux"]),
         (M, 768, 768, ["bias", "resid", "resid_map"]), (M, 768, 3072, ["bias", "resid", "resid_mask"]),
         (M, 768, 2304, ["bias"]), (25088, 8192, 768, ["bias"]), (4096, 512, 256, ["bias", "gelu", "resid"]),
         (8192, 256, 128, ["bias"])]
if quick:
    cases = [(8192, 768, 768, ["bias", "bias_scale", "gelu", "gelu_dg", "dgelu", "mul_aux", "resid", "resid_map", "resid_mask"]),
             (4096, 256, 128, ["bias", "gelu", "resid"])]
out = []
for (m, n, k, epis) in cases:
    torch.manual_seed(m + n + k)
    A = torch.randn(m, k, device="cuda").bfloat16()
    B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(n, device="cuda")
    gamma = torch.randn(n, device="cuda") * 0.1
    aux16 = torch.randn(m, n, device="cuda").bfloat16()
    nb = m // T + 2
    x_in = torch.randn((nb + 1) * T + 256, n, device="cuda")
    keep = (torch.arange(nb + 2, device="cuda") % 5 != 0).float()
    smap = torch.cat([torch.randperm(nb + 2, device="cuda"), torch.zeros(256, dtype=torch.int64, device="cuda")]).to(torch.int32)
    for epi in epis:
        res = {}
        for mode in (0, 1):
            _lib.set_option("gemm_p8d", mode)
            o0 = torch.full((m, n), 3.0, dtype=torch.bfloat16, device="cuda")
            o1 = torch.full((m, n), 3.0, dtype=torch.bfloat16, device="cuda")
            cs = torch.zeros(8, n, device="cuda")
            xr = torch.full_like(x_in, 7.0)
            if epi == "bias":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o0, bias=bias)
            elif epi == "bias_scale":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o0, bias=bias, colscale=0.125, colscale_n=n // 3, colsum=cs, colsum_copies=8)
            elif epi == "gelu":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU, out0=o0, out1=o1, bias=bias)
            elif epi == "gelu_dg":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_GELU_DG, out0=o0, out1=o1, bias=bias)
            elif epi == "dgelu":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_DGELU, out0=o0, aux=aux16, colsum=cs, colsum_copies=8)
            elif epi == "mul_aux":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_MUL_AUX, out0=o0, aux=aux16, colsum=cs, colsum_copies=8)
            elif epi == "resid":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_RESIDUAL, bias=bias, vec1=gamma, resid=xr, aux=x_in, ldaux=n, rows_per_sample=T)
            elif epi == "resid_mask":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_RESIDUAL, bias=bias, vec1=gamma, resid=xr, aux=x_in, ldaux=n, rowmask=keep, keep_prob=0.9, rows_per_sample=T)
            elif epi == "resid_map":
                f = lambda: ops.gemm_nt(A, B, m, n, k, ops.EPI_RESIDUAL, bias=bias, vec1=gamma, resid=xr, aux=x_in, ldaux=n, keep_prob=0.9, rows_per_sample=T, sample_map=smap)
            f()
            torch.cuda.synchronize()
            snap = [o0.clone(), o1.clone(), xr.clone()]
            cs1 = cs.clone()
            t = timed(f)
            res[mode] = (snap, cs1, t)
        _lib.set_option("gemm_p8d", 1)
        eq = [bool(torch.equal(a, b)) for a, b in zip(res[0][0], res[1][0])]
        csd = float((res[0][1].sum(0) - res[1][1].sum(0)).abs().max())
        fl = 2.0 * m * n * k
        line = dict(M=m, N=n, K=k, epi=epi, equal=eq, colsum_maxdiff=csd, p8_us=round(res[0][2], 1), p8d_us=round(res[1][2], 1),
                    p8_tf=round(fl / res[0][2] / 1e6), p8d_tf=round(fl / res[1][2] / 1e6))
        out.append(line)
        print(json.dumps(line), flush=True)
bad = [l for l in out if not all(l["equal"])]
print("MISMATCHES:", len(bad))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/p8d_check.json", "w"), indent=1)
sys.exit(1 if bad else 0)
