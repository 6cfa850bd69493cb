"""-m gpu: bf16 MFMA GEMM + fused epilogues vs plain torch fp32 references of the same op."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale)


def test_exact_integer_layout():
    """A = I-like / small integers, asymmetric B: every product is exact in bf16/fp32, so any
    fragment-layout or swizzle mistake shows up as a hard mismatch (not a tolerance issue)."""
    from mem_amd import ops
    M, N, K = 256, 384, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    A = torch.randint(-3, 4, (M, K), generator=g, device="cuda").float()
    B = torch.randint(-3, 4, (N, K), generator=g, device="cuda").float()
    B += torch.arange(N, device="cuda").view(N, 1) % 5            # asymmetric
    C = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    ops.gemm_nt(A.bfloat16(), B.bfloat16(), M, N, K, ops.EPI_F32, out0=C)
    torch.testing.assert_close(C, A @ B.t(), rtol=0, atol=0)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (394, 768, 768), (50, 200, 64), (1000, 2304, 768),
                                   (333, 3072, 768), (394, 768, 3072), (196 * 3, 128, 768), (7, 512, 128)])
def test_bias_bf16_and_edges(M, N, K):
    from mem_amd import ops
    A = _rand((M, K), 1).bfloat16()
    B = _rand((N, K), 2, 0.05).bfloat16()
    bias = _rand((N,), 3)
    out = torch.full((M + 3, N), 7.0, dtype=torch.bfloat16, device="cuda")       # guard rows
    ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_BF16, out0=out, bias=bias, colscale=0.125, colscale_n=N // 3)
    ref = (A.float() @ B.float().t() + bias).bfloat16()
    ref[:, : N // 3] = (ref[:, : N // 3].float() * 0.125).bfloat16()
    torch.testing.assert_close(out[:M].float(), ref.float(), rtol=2e-2, atol=2e-2)
    # fp32-accumulate quality: compare against the fp64 product at bf16 resolution
    ref64 = (A.double() @ B.double().t() + bias.double())
    ref64[:, : N // 3] *= 0.125
    err = (out[:M].double() - ref64).abs().max().item()
    assert err < 0.02 * max(1.0, ref64.abs().max().item())
    assert (out[M:] == 7.0).all()                                                # rows >= M untouched


def test_gelu_epilogue():
    from mem_amd import ops
    M, N, K = 300, 512, 256
    A, B, bias = _rand((M, K), 4).bfloat16(), _rand((N, K), 5, 0.08).bfloat16(), _rand((N,), 6, 0.1)
    h = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    a = torch.zeros_like(h)
    ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_GELU, out0=h, out1=a, bias=bias)
    href = (A.float() @ B.float().t() + bias).bfloat16()
    torch.testing.assert_close(h.float(), href.float(), rtol=2e-2, atol=2e-2)
    # GELU is applied to the kernel's own rounded h: must match torch's exact-erf GELU on that h
    torch.testing.assert_close(a.float(), torch.nn.functional.gelu(h.float()).bfloat16().float(), rtol=1e-2, atol=1e-3)


def test_residual_epilogue_with_droppath():
    from mem_amd import ops
    T, Bn, N, K = 17, 9, 256, 128
    M = T * Bn
    A, B, bias = _rand((M, K), 7).bfloat16(), _rand((N, K), 8, 0.08).bfloat16(), _rand((N,), 9, 0.1)
    gamma = _rand((N,), 10, 0.1)
    x0 = _rand((M, N), 11)
    keep = (torch.arange(Bn, device="cuda") % 3 != 0).float()
    for mask, kp in ((None, 1.0), (keep, 0.9)):
        x = x0.clone()
        y = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        ops.gemm_nt(A, B, M, N, K, ops.EPI_RESIDUAL, out0=y, bias=bias, vec1=gamma, resid=x, rowmask=mask,
                    keep_prob=kp, rows_per_sample=T)
        t = gamma * y.float()
        if mask is not None:
            # true division like the CPU reference (torch-GPU multiplies by a reciprocal instead)
            t = (t.cpu().div(kp) * mask.cpu().repeat_interleave(T).view(-1, 1)).cuda()
        torch.testing.assert_close(x, x0 + t, rtol=0, atol=0)          # fp32 tail is exact given y
        torch.testing.assert_close(y.float(), (A.float() @ B.float().t() + bias).bfloat16().float(), rtol=2e-2, atol=2e-2)


def test_dgelu_and_f32_accumulate():
    from mem_amd import ops
    M, N, K = 260, 384, 192
    A, B = _rand((M, K), 12).bfloat16(), _rand((N, K), 13, 0.08).bfloat16()
    h = _rand((M, N), 14).bfloat16()
    out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros(N, device="cuda")
    ops.gemm_nt(A, B, M, N, K, ops.EPI_DGELU, out0=out, aux=h, colsum=cs)
    torch.testing.assert_close(cs, out.float().sum(0), rtol=1e-3, atol=5e-2)
    da = (A.float() @ B.float().t()).bfloat16().float()
    hh = h.float().requires_grad_(True)
    torch.nn.functional.gelu(hh).backward(torch.ones_like(hh))
    ref = (da * hh.grad).bfloat16().float()
    torch.testing.assert_close(out.float(), ref, rtol=3e-2, atol=3e-2)
    C = torch.ones((M, N), dtype=torch.float32, device="cuda")
    ops.gemm_nt(A, B, M, N, K, ops.EPI_F32, out0=C, accumulate=True)
    torch.testing.assert_close(C, 1.0 + A.float() @ B.float().t(), rtol=1e-4, atol=1e-3)


def test_patch_embed_epilogue():
    from mem_amd import ops
    Bn, L, N, K = 5, 16, 128, 768
    M = Bn * L
    A, Wt, bias = _rand((M, K), 15).bfloat16(), _rand((N, K), 16, 0.05).bfloat16(), _rand((N,), 17, 0.1)
    mt = _rand((N,), 18)
    mask = (torch.rand((M,), device="cuda") < 0.4).to(torch.uint8)
    x = torch.full((Bn * (L + 1), N), -5.0, device="cuda")
    ops.gemm_nt(A, Wt, M, N, K, ops.EPI_PATCH_EMBED, bias=bias, vec1=mt, resid=x, aux=mask, rows_per_sample=L)
    y = (A.float() @ Wt.float().t() + bias).bfloat16().float()
    w = mask.float().view(-1, 1)
    ref = (y * (1 - w) + mt * w).view(Bn, L, N)
    got = x.view(Bn, L + 1, N)
    torch.testing.assert_close(got[:, 1:], ref, rtol=2e-2, atol=2e-2)
    assert (got[:, 0] == -5.0).all()                                  # cls rows untouched


def test_vitb_shapes_throughput_sanity():
    """Full BASELINE M (256*197) on the three ViT-B weight shapes: finite + spot-check rows."""
    from mem_amd import ops
    M = 256 * 197
    for N, K in ((2304, 768), (3072, 768), (768, 3072)):
        A, B = _rand((M, K), 20).bfloat16(), _rand((N, K), 21, 0.03).bfloat16()
        out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_BF16, out0=out)
        rows = torch.tensor([0, 1, 127, 128, 25000, M - 1], device="cuda")
        ref = (A[rows].float() @ B.float().t()).bfloat16().float()
        torch.testing.assert_close(out[rows].float(), ref, rtol=2e-2, atol=2e-2)
        assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("R,N,K", [(256, 128, 128), (394, 768, 768), (50432 // 8, 2304, 768), (1000, 512, 3072),
                                   (333, 128, 512), (70, 768, 512), (5000, 8192, 768), (2048, 256, 256),
                                   (50432, 768, 768), (4099, 3072, 768), (3000, 768, 3072), (2500, 512, 256)])
def test_gemm_tn_weight_gradient(R, N, K):
    """out[N,K] += A[R,N]^T @ B[R,K] with the transposing LDS read + split-K atomics."""
    from mem_amd import ops
    # exact-integer check first (layout / k-permutation / swizzle mistakes are hard failures)
    g = torch.Generator(device="cuda").manual_seed(R)
    Ai = torch.randint(-2, 3, (R, N), generator=g, device="cuda").float()
    Bi = torch.randint(-2, 3, (R, K), generator=g, device="cuda").float() + (torch.arange(K, device="cuda") % 3)
    out = torch.zeros((N, K), dtype=torch.float32, device="cuda")
    ops.gemm_tn(Ai.bfloat16(), Bi.bfloat16(), R, N, K, out, accumulate=True)
    torch.testing.assert_close(out, Ai.t() @ Bi, rtol=0, atol=0)
    A, B = _rand((R, N), 50).bfloat16(), _rand((R, K), 51).bfloat16()
    out = torch.full((N, K), 2.0, dtype=torch.float32, device="cuda")
    ops.gemm_tn(A, B, R, N, K, out, accumulate=True)
    ref = 2.0 + A.float().t() @ B.float()
    torch.testing.assert_close(out, ref, rtol=2e-3, atol=2e-2)
    out2 = torch.full((N, K), 5.0, dtype=torch.float32, device="cuda")
    ops.gemm_tn(A, B, R, N, K, out2, accumulate=False)
    torch.testing.assert_close(out2, ref - 2.0, rtol=2e-3, atol=2e-2)
    # workspace form: partial tiles by plain stores + a reduction pass; fixed sum order => deterministic
    need = ops.gemm_tn_workspace(R, N, K)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    o1 = torch.full((N, K), -5.0, device="cuda")
    ops.gemm_tn(A, B, R, N, K, o1, accumulate=False, workspace=ws)
    torch.testing.assert_close(o1, ref - 2.0, rtol=2e-3, atol=2e-2)
    o2 = torch.full((N, K), 9.0, device="cuda")
    ops.gemm_tn(A, B, R, N, K, o2, accumulate=False, workspace=ws)
    if need:
        assert torch.equal(o1, o2)
    ops.gemm_tn(A, B, R, N, K, o2, accumulate=True, workspace=ws)
    torch.testing.assert_close(o2, 2 * (ref - 2.0), rtol=2e-3, atol=4e-2)
    cs = torch.zeros(N, device="cuda")
    ops.colsum_bf16(A, R, N, cs)
    torch.testing.assert_close(cs, A.float().sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("shapes", [
    [(50432, 768, 768), (50432, 2304, 768)],                       # proj + qkv of a ViT-B block: one grid, 7 row slices each
    [(9001, 256, 512), (5000, 512, 256), (3000, 256, 256)],        # different row counts, ragged
    [(4099, 768, 768), (4099, 768, 3072), (4099, 3072, 768), (4099, 2304, 768)],
    [(3000, 768, 768), (300, 768, 768)],                           # second product outside the tile kernel: one by one
    [(2500, 512, 256)],                                            # a group of one
])
def test_gemm_tn_group_equals_single_calls(shapes):
    """memhip_gemm_bf16_tn_group: every product of the group has the single call's contract (exact on integer data, equal to
    the single call to fp32 rounding -- the row slices differ --, accumulate adds, same bits every run), whether the group
    runs as one grid or falls back to the single calls."""
    from mem_amd import ops
    g = torch.Generator(device="cuda").manual_seed(len(shapes))
    ints = [(torch.randint(-2, 3, (R, N), generator=g, device="cuda").float(),
             torch.randint(-2, 3, (R, K), generator=g, device="cuda").float() + (torch.arange(K, device="cuda") % 3)) for R, N, K in shapes]
    need = max(ops.gemm_tn_group_workspace(shapes), 16)
    assert need >= max(ops.gemm_tn_workspace(*sh) for sh in shapes)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    outs = [torch.full((N, K), 3.0, device="cuda") for _, N, K in shapes]
    ops.gemm_tn_group([(A.bfloat16(), B.bfloat16(), R, N, K, o) for (A, B), (R, N, K), o in zip(ints, shapes, outs)],
                      accumulate=False, workspace=ws)
    for (A, B), o in zip(ints, outs):
        torch.testing.assert_close(o, A.t() @ B, rtol=0, atol=0)
    ops_in = [(_rand((R, N), 60 + i).bfloat16(), _rand((R, K), 70 + i).bfloat16()) for i, (R, N, K) in enumerate(shapes)]
    single = [torch.zeros((N, K), device="cuda") for _, N, K in shapes]
    for (A, B), (R, N, K), o in zip(ops_in, shapes, single):
        ops.gemm_tn(A, B, R, N, K, o, accumulate=False, workspace=ws)
    probs = [(A, B, R, N, K, o) for (A, B), (R, N, K), o in zip(ops_in, shapes, outs)]
    ops.gemm_tn_group(probs, accumulate=False, workspace=ws)
    first = [o.clone() for o in outs]
    for o, s1 in zip(outs, single):
        torch.testing.assert_close(o, s1, rtol=1e-5, atol=1e-5 * float(s1.abs().max()))
    ops.gemm_tn_group(probs, accumulate=False, workspace=ws)
    assert all(torch.equal(o, f) for o, f in zip(outs, first))
    ops.gemm_tn_group(probs, accumulate=True, workspace=ws)
    for o, f in zip(outs, first):
        torch.testing.assert_close(o, 2 * f, rtol=1e-6, atol=0)
    # without a workspace the products run one by one (atomics): same values to rounding
    ops.gemm_tn_group(probs, accumulate=False, workspace=None)
    for o, s1 in zip(outs, single):
        torch.testing.assert_close(o, s1, rtol=1e-4, atol=1e-5 * float(s1.abs().max()))


@pytest.mark.parametrize("M,N,K", [(2048, 128, 64), (4000, 768, 768), (2304 + 17, 2304, 768), (5000, 384, 3072),
                                   (4096, 256, 64), (4096 + 100, 768, 768), (9000, 2304, 128), (4500, 512, 3072),
                                   (25600 + 13, 768, 1536)])       # last: 303 tiles -> row-split launch (p8 + 128x128)
def test_persistent_gemm_large_m_all_epilogues(M, N, K):
    """Large-M shapes dispatch to the persistent 256-row-tile kernels (gemm_p8.hip, row-split launches with 128-row
    tiles for the left-over rows; gemm256.hip when K is not a multiple of 128; the 128x128 kernel otherwise):
    exact-integer layout check + every fused epilogue against torch."""
    from mem_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N)
    Ai = torch.randint(-3, 4, (M, K), generator=g, device="cuda").float()
    Bi = torch.randint(-3, 4, (N, K), generator=g, device="cuda").float() + (torch.arange(N, device="cuda").view(N, 1) % 5)
    C = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    ops.gemm_nt(Ai.bfloat16(), Bi.bfloat16(), M, N, K, ops.EPI_F32, out0=C)
    torch.testing.assert_close(C, Ai @ Bi.t(), rtol=0, atol=0)
    ops.gemm_nt(Ai.bfloat16(), Bi.bfloat16(), M, N, K, ops.EPI_F32, out0=C, accumulate=True)
    torch.testing.assert_close(C, 2 * (Ai @ Bi.t()), rtol=0, atol=0)
    A, B, bias = _rand((M, K), 1).bfloat16(), _rand((N, K), 2, 0.05).bfloat16(), _rand((N,), 3)
    ref = A.float() @ B.float().t() + bias
    out = torch.full((M + 5, N), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_BF16, out0=out, bias=bias, colscale=0.125, colscale_n=N // 2)
    r = ref.bfloat16()
    r[:, : N // 2] = (r[:, : N // 2].float() * 0.125).bfloat16()
    torch.testing.assert_close(out[:M].float(), r.float(), rtol=2e-2, atol=2e-2)
    assert (out[M:] == 7.0).all()
    h, a = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda"), torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_GELU, out0=h, out1=a, bias=bias)
    torch.testing.assert_close(h.float(), ref.bfloat16().float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(a.float(), torch.nn.functional.gelu(h.float()).bfloat16().float(), rtol=1e-2, atol=1e-3)
    T = 17
    Bn = (M + T - 1) // T
    gamma, x0 = _rand((N,), 10, 0.1), _rand((M, N), 11)
    keep = (torch.arange(Bn, device="cuda") % 3 != 0).float()
    x, y = torch.empty_like(x0), torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(A, B, M, N, K, ops.EPI_RESIDUAL, out0=y, bias=bias, vec1=gamma, resid=x, aux=x0, ldaux=N, rowmask=keep,
                keep_prob=0.9, rows_per_sample=T)
    t = (gamma * y.float()).cpu().div(0.9) * keep.cpu().repeat_interleave(T)[:M].view(-1, 1)
    torch.testing.assert_close(x, x0 + t.cuda(), rtol=0, atol=0)
    torch.testing.assert_close(y.float(), ref.bfloat16().float(), rtol=2e-2, atol=2e-2)
    hh = _rand((M, N), 14).bfloat16()
    o = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    cs = torch.zeros(N, device="cuda")
    ops.gemm_nt(A, B, M, N, K, ops.EPI_DGELU, out0=o, aux=hh, colsum=cs)
    da = (A.float() @ B.float().t()).bfloat16().float()
    hg = hh.float().requires_grad_(True)
    torch.nn.functional.gelu(hg).backward(torch.ones_like(hg))
    torch.testing.assert_close(o.float(), (da * hg.grad).bfloat16().float(), rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(cs, o.float().sum(0), rtol=1e-3, atol=5e-2)      # fused bias-gradient column sums
    # the same sums through accumulator copies (colsum_copies) + fold: equal to the single accumulator up to fp32
    # summation order; the workspace is zero again afterwards and the fold ADDS into its target
    for copies in (8, 3):
        ws = torch.zeros(copies, N, device="cuda")
        o2 = torch.zeros_like(o)
        tgt = torch.full((N,), 2.0, device="cuda")
        ops.gemm_nt(A, B, M, N, K, ops.EPI_DGELU, out0=o2, aux=hh, colsum=ws, colsum_copies=copies)
        ops.colsum_fold(ws, copies, N, tgt)
        assert torch.equal(o2, o)
        torch.testing.assert_close(tgt - 2.0, cs, rtol=1e-4, atol=1e-2)
        assert float(ws.abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(300, 256, 128), (4096 + 37, 1024, 256), (5000, 768, 768)])
def test_gelu_with_stored_derivative(M, N, K):
    """BIAS_GELU_DG stores gelu'(h) as fp16 (16 bits per value, 11 significant) next to gelu(h); MUL_AUX multiplies the
    incoming gradient with it: together they reproduce the DGELU path (gelu' in fp32 from the stored h) up to the fp16
    rounding of gelu' -- a relative 2^-11 in front of the final bf16 rounding."""
    from mem_amd import ops
    A, B, bias = _rand((M, K), 1).bfloat16(), _rand((N, K), 2, 0.05).bfloat16(), _rand((N,), 3)
    h = (A.float() @ B.float().t() + bias).bfloat16().float()
    dg, a = torch.zeros((M, N), dtype=torch.float16, device="cuda"), torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_GELU_DG, out0=dg, out1=a, bias=bias)
    hg = h.clone().requires_grad_(True)
    torch.nn.functional.gelu(hg).backward(torch.ones_like(hg))
    torch.testing.assert_close(a.float(), torch.nn.functional.gelu(h).bfloat16().float(), rtol=2e-2, atol=2e-3)
    # fp16 resolution: 2^-11 relative (+ the 5e-7 of the erf approximation); bf16 would be 2^-8.  (h of the kernel and h
    # of this reference can differ by one bf16 ulp where the fp32 sums differ: compare where they agree)
    hk = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(A, B, M, N, K, ops.EPI_BIAS_GELU, out0=hk, out1=torch.empty_like(hk), bias=bias)
    same = hk.float() == h
    assert float(same.float().mean()) > 0.95
    err = (dg.float() - hg.grad).abs()[same]
    assert float(err.max()) <= 6e-4, float(err.max())
    G, W2 = _rand((M, K), 7).bfloat16(), _rand((N, K), 8, 0.05).bfloat16()
    o, cs = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda"), torch.zeros(N, device="cuda")
    ops.gemm_nt(G, W2, M, N, K, ops.EPI_MUL_AUX, out0=o, aux=dg, colsum=cs)
    da = (G.float() @ W2.float().t()).bfloat16().float()
    torch.testing.assert_close(o.float(), (da * dg.float()).bfloat16().float(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(cs, o.float().sum(0), rtol=1e-3, atol=5e-2)
    # against the DGELU path on the same h: equal except where the fp16 rounding of gelu' moves the product across a bf16
    # rounding boundary (expected ~6 % of the elements, by one bf16 ulp)
    o2 = torch.zeros_like(o)
    ops.gemm_nt(G, W2, M, N, K, ops.EPI_DGELU, out0=o2, aux=hk)
    diff = (o.float() - o2.float()).abs()
    # (floor: gelu' below fp16's subnormal spacing of 6e-8 is flushed; times |da| of a few units)
    ulp = torch.maximum(o2.float().abs() * 2.0 ** -7, torch.tensor(1e-5, device="cuda"))
    assert float((diff > 0).float().mean()) < 0.12
    assert bool((diff <= ulp).all())


def test_gemm_dispatch_fuzz_exact():
    """Randomised shapes across every dispatch boundary (128x128 kernel, gemm256, gemm_p8 full launches, row-split
    launches with 128-row tiles, weight-gradient kernels with and without the workspace): small-integer operands, so
    fp32 accumulation is exact and any mis-addressed tile / dropped row / double-counted split shows as a bit
    difference."""
    from mem_amd import ops
    rng = np.random.default_rng(123)
    g = torch.Generator(device="cuda").manual_seed(9)
    shapes = []
    for _ in range(14):
        shapes.append((int(rng.integers(1, 9000)), 8 * int(rng.integers(1, 200)), 64 * int(rng.integers(1, 20))))
    shapes += [(4096, 256, 128), (4095, 1024, 192), (12289, 1024, 1024), (19216, 4096, 1024), (19216, 1024, 4096),
               (8193, 768, 64), (50432, 256, 128), (4097, 3072, 320)]
    for (M, N, K) in shapes:
        A = torch.randint(-2, 3, (M, K), generator=g, device="cuda").float()
        B = torch.randint(-2, 3, (N, K), generator=g, device="cuda").float()
        C = torch.full((M + 3, N), 5.0, dtype=torch.float32, device="cuda")
        ops.gemm_nt(A.bfloat16(), B.bfloat16(), M, N, K, ops.EPI_F32, out0=C)
        assert torch.equal(C[:M], A @ B.t()), (M, N, K)
        assert (C[M:] == 5.0).all(), (M, N, K)
    for _ in range(10):
        R, N, K = int(rng.integers(1, 9000)), 8 * int(rng.integers(1, 160)), 8 * int(rng.integers(1, 160))
        dY = torch.randint(-2, 3, (R, N), generator=g, device="cuda").float()
        X = torch.randint(-2, 3, (R, K), generator=g, device="cuda").float()
        want = dY.t() @ X
        out = torch.ones((N, K), dtype=torch.float32, device="cuda")
        ops.gemm_tn(dY.bfloat16(), X.bfloat16(), R, N, K, out, accumulate=True)
        assert torch.equal(out, want + 1.0), (R, N, K)
        ws = torch.empty(max(ops.gemm_tn_workspace(R, N, K), 16), dtype=torch.uint8, device="cuda")
        ops.gemm_tn(dY.bfloat16(), X.bfloat16(), R, N, K, out, accumulate=False, workspace=ws)
        assert torch.equal(out, want), (R, N, K)
