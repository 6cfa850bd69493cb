"""dVAE tokenizer forward on the HIP path vs the reference's fp32 outputs
(reference: eventvae/vae/vae_model.py:29-113,153-158; fp32 because mem/engine_for_pretraining.py:140-145 is outside
the autocast block).  Labels are integers: the DEFAULT mode (fp32 MFMA, csrc/conv_f32.hip) must EQUAL the reference ids
(tests/golden/vae_tiny.npz, vae_base.npz = the ViT-B tokenizer shape); the opt-in bf16 mode (csrc/conv.hip) is held to
token agreement with disagreements only at near ties.  Plus layer-level numeric checks of both convolutions."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _vae(hidden, tokens, res, size, layers=4, seed=0):
    from mem_amd.vae_model import DiscreteVAE
    torch.manual_seed(seed)
    return DiscreteVAE(input_H=size, input_W=size, num_tokens=tokens, codebook_dim=64, num_layers=layers,
                       num_resnet_blocks=res, hidden_dim=hidden, channels=3).cuda().eval()


@pytest.mark.parametrize("k,s,p,cin,cout,h", [(4, 2, 1, 64, 128, 16), (3, 1, 1, 128, 64, 14), (1, 1, 0, 64, 256, 14),
                                              (4, 2, 1, 4, 64, 32)])
def test_conv_layers_vs_torch(k, s, p, cin, cout, h):
    from mem_amd import ops
    g = torch.Generator(device="cuda").manual_seed(k * 100 + cin)
    B = 3
    x = torch.randn(B, cin, h, h, generator=g, device="cuda")
    if cin == 4:
        x[:, 3] = 0
    w = torch.randn(cout, cin, k, k, generator=g, device="cuda") * 0.05
    b = torch.randn(cout, generator=g, device="cuda")
    xb, wb = x.bfloat16(), w.bfloat16()
    ref = torch.nn.functional.conv2d(xb.float(), wb.float(), b, stride=s, padding=p)
    ho = ref.shape[-1]
    xp = torch.zeros(B, h + 2, h + 2, cin, dtype=torch.bfloat16, device="cuda")
    xp[:, 1:-1, 1:-1] = xb.permute(0, 2, 3, 1)
    wp = wb.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = torch.zeros(B, ho + 2, ho + 2, cout, dtype=torch.bfloat16, device="cuda")
    add = torch.zeros_like(out)
    add[:, 1:-1, 1:-1] = torch.randn(B, ho, ho, cout, generator=g, device="cuda").bfloat16()
    ops.conv2d_nhwc(xp, wp, b, out, B, h, h, cin, cout, k, s, p, relu=True)
    got = out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
    torch.testing.assert_close(got, torch.relu(ref).bfloat16().float(), rtol=2e-2, atol=2e-2)
    assert out[:, 0].abs().max() == 0 and out[:, :, 0].abs().max() == 0          # the border stays zero
    ops.conv2d_nhwc(xp, wp, b, out, B, h, h, cin, cout, k, s, p, relu=False, add=add)
    want = (ref.bfloat16().float() + add[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()).bfloat16().float()
    torch.testing.assert_close(out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float(), want, rtol=2e-2, atol=3e-2)
    dense = torch.zeros(B * ho * ho, cout, dtype=torch.bfloat16, device="cuda")
    ops.conv2d_nhwc(xp, wp, b, dense, B, h, h, cin, cout, k, s, p, relu=False, out_padded=False)
    torch.testing.assert_close(dense.view(B, ho, ho, cout).permute(0, 3, 1, 2).float(), ref.bfloat16().float(),
                               rtol=2e-2, atol=2e-2)


def test_argmax_rows_first_maximum():
    from mem_amd import ops
    x = torch.randn(1000, 8192, device="cuda").bfloat16()
    x[5, 100] = x[5, 7000] = 50.0                                  # tie: first index wins
    ids = torch.empty(1000, dtype=torch.int64, device="cuda")
    ops.argmax_rows(x, 1000, 8192, ids)
    assert ids[5].item() == 100
    assert torch.equal(x.float().gather(1, ids.view(-1, 1)).view(-1), x.float().max(1).values)


@pytest.mark.parametrize("k,s,p,cin,cout,h", [(4, 2, 1, 64, 128, 16), (3, 1, 1, 128, 64, 14), (1, 1, 0, 64, 256, 14),
                                              (4, 2, 1, 4, 64, 32), (3, 1, 1, 384, 384, 14), (1, 1, 0, 384, 520, 7),
                                              (2, 2, 0, 8, 12, 10)])
def test_conv_f32_layers_vs_torch(k, s, p, cin, cout, h):
    """fp32 implicit GEMM (v_mfma_f32_16x16x4_f32) vs torch's fp32 convolution in float64: relative error at the fp32
    accumulation level (<= 4e-7 sqrt(K) of the output scale), ragged M / N tiles, bias + ReLU + residual, dense output."""
    from mem_amd import ops
    g = torch.Generator(device="cuda").manual_seed(k * 100 + cin)
    B = 3
    x = torch.randn(B, cin, h, h, generator=g, device="cuda")
    if cin == 4:
        x[:, 3] = 0
    w = torch.randn(cout, cin, k, k, generator=g, device="cuda") * 0.05
    b = torch.randn(cout, generator=g, device="cuda")
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p)
    ho = ref.shape[-1]
    xp = torch.zeros(B, h + 2, h + 2, cin, device="cuda")
    xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    wp = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous()
    out = torch.zeros(B, ho + 2, ho + 2, cout, device="cuda")
    add = torch.zeros_like(out)
    add[:, 1:-1, 1:-1] = torch.randn(B, ho, ho, cout, generator=g, device="cuda")
    # fp32 accumulation over K = k*k*cin terms: error ~ sqrt(K) * 2^-24 * scale (measured 1.1e-6 at K = 3456)
    tol = 4e-7 * max(4.0, (k * k * cin) ** 0.5) * float(ref.abs().max())
    ops.conv2d_nhwc(xp, wp, b, out, B, h, h, cin, cout, k, s, p, relu=True)
    assert (out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double() - torch.relu(ref)).abs().max().item() <= tol
    assert out[:, 0].abs().max() == 0 and out[:, :, 0].abs().max() == 0 and out[:, -1].abs().max() == 0
    ops.conv2d_nhwc(xp, wp, b, out, B, h, h, cin, cout, k, s, p, relu=False, add=add)
    want = ref + add[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double()
    assert (out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double() - want).abs().max().item() <= tol
    dense = torch.zeros(B * ho * ho, cout, device="cuda")
    ops.conv2d_nhwc(xp, wp, None, dense, B, h, h, cin, cout, k, s, p, relu=False, out_padded=False)
    want = ref - b.double().view(1, -1, 1, 1)
    assert (dense.view(B, ho, ho, cout).permute(0, 3, 1, 2).double() - want).abs().max().item() <= tol


def test_argmax_rows_f32_first_maximum_and_gap():
    from mem_amd import ops
    x = torch.randn(1000, 8192, device="cuda")
    x[5, 100] = x[5, 7000] = 50.0                                  # tie: first index wins, gap 0
    x[6, 8191] = 60.0
    ids = torch.empty(1000, dtype=torch.int64, device="cuda")
    gap = torch.empty(1000, device="cuda")
    ops.argmax_rows(x, 1000, 8192, ids, gap)
    assert ids[5].item() == 100 and gap[5].item() == 0.0 and ids[6].item() == 8191
    assert torch.equal(ids, x.argmax(1))
    top2 = x.topk(2, dim=1).values
    assert torch.equal(gap, top2[:, 0] - top2[:, 1])


def test_argmax_rows_nan_and_all_inf_rows_give_valid_ids():
    """torch.argmax: NaN beats every number (first NaN), a row of all -inf gives 0.  The ids are used unchecked as
    labels, so they must always be in range (a non-finite input image used to yield id 0x7fffffff)."""
    from mem_amd import ops
    x = torch.randn(64, 8192, device="cuda")
    x[3] = float("nan")
    x[4] = float("-inf")
    x[5, 4000] = float("nan"); x[5, 77] = 1e30
    x[6, 8000] = float("nan"); x[6, 300] = float("nan")
    x[7, :5000] = float("-inf")
    ids = torch.empty(64, dtype=torch.int64, device="cuda")
    ops.argmax_rows(x, 64, 8192, ids)
    assert torch.equal(ids, x.argmax(1)) and ids[3].item() == 0 and ids[4].item() == 0 and ids[5].item() == 4000
    assert ids[6].item() == 300
    xb = x.bfloat16()
    ops.argmax_rows(xb, 64, 8192, ids)
    assert int(ids.min()) >= 0 and int(ids.max()) < 8192
    assert ids[3].item() == 0 and ids[4].item() == 0 and ids[5].item() == 4000 and ids[6].item() == 300


def test_tokenizer_fp32_equals_reference_golden():
    """ids written by the REFERENCE DiscreteVAE in fp32 (oracle/gen_golden_vae.py): the default HIP mode must reproduce
    every one of them -- tiny config and the ViT-B tokenizer shape (hidden 384, 3 ResBlocks, 8192 tokens, 224^2)."""
    import os
    import numpy as np
    from oracle.vae_ref import BASE_VAE, TINY_VAE, fill_vae_by_name, vae_inputs
    from mem_amd.vae_model import DiscreteVAE, HipTokenizer
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gdir, "vae_tiny.npz"))
    m = DiscreteVAE(**TINY_VAE).eval()
    m.load_state_dict(fill_vae_by_name(m.state_dict(), seed=0))
    tok = HipTokenizer(m.cuda(), max_batch=6)
    assert tok.precision == "fp32"
    ids = tok.get_codebook_indices(vae_inputs(TINY_VAE, 6, 11).cuda()).cpu().numpy()
    assert np.array_equal(ids, g["ids"])
    assert np.abs(tok.last_top2_gap(6).cpu().numpy() - g["top2_gap"]).max() <= 1e-4
    g = np.load(os.path.join(gdir, "vae_base.npz"))
    m = DiscreteVAE(**BASE_VAE).eval()
    m.load_state_dict(fill_vae_by_name(m.state_dict(), seed=1))
    tok = HipTokenizer(m.cuda(), max_batch=2)
    img = vae_inputs(BASE_VAE, 2, 12) * (vae_inputs(BASE_VAE, 2, 13) < 0.3)
    ids = tok.get_codebook_indices(img.cuda()).cpu().numpy()
    assert ids.shape == (2, 196) and np.array_equal(ids, g["ids"])
    assert np.abs(tok.last_top2_gap(2).cpu().numpy() - g["top2_gap"]).max() <= 1e-4 * float(g["logit_std"]) + 1e-5


@pytest.mark.parametrize("hidden,tokens,res,size", [(64, 512, 2, 64), (128, 1024, 1, 96)])
def test_tokenizer_fp32_equals_fp64_module(hidden, tokens, res, size):
    """Random weights, 8 images: ids equal the float64 evaluation of the same module wherever its top-2 gap exceeds
    fp32 accumulation noise (1e-4 of the logit spread)."""
    from mem_amd.vae_model import HipTokenizer
    vae = _vae(hidden, tokens, res, size)
    tok = HipTokenizer(vae, max_batch=8)
    img = torch.rand(8, 3, size, size, device="cuda")
    ids = tok.get_codebook_indices(img)
    lg = vae.double()(img.double(), return_logits=True).flatten(2).transpose(1, 2)
    vae.float()
    ref = lg.argmax(-1)
    top2 = lg.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-4 * lg.std()
    assert safe.float().mean().item() > 0.99
    assert torch.equal(ids[safe], ref[safe])


@pytest.mark.parametrize("hidden,tokens,res,size", [(64, 512, 2, 64), (128, 1024, 1, 96)])
def test_tokenizer_agreement_with_fp32_module(hidden, tokens, res, size):
    from mem_amd.vae_model import HipTokenizer
    vae = _vae(hidden, tokens, res, size)
    tok = HipTokenizer(vae, max_batch=8, precision="bf16")
    img = torch.rand(8, 3, size, size, device="cuda")
    ref = vae.get_codebook_indices(img)
    ids = tok.get_codebook_indices(img)
    assert ids.shape == ref.shape and ids.dtype == torch.int64
    agree = (ids == ref).float().mean().item()
    # where they differ, the fp32 logits of the two candidates must be a near tie
    lg = vae(img, return_logits=True).flatten(2).transpose(1, 2)                  # [B, hw, tokens]
    gap = (lg.gather(2, ref.unsqueeze(-1)) - lg.gather(2, ids.unsqueeze(-1))).squeeze(-1)
    spread = lg.std().item()
    assert agree >= 0.97, agree
    assert gap.max().item() <= 0.05 * spread + 1e-3, (gap.max().item(), spread)


def test_tokenizer_vit_b_config_shapes():
    """The MEM tokenizer (4 layers, hidden 384, 3 ResBlocks, 8192 tokens, 224x224) at a small batch."""
    from mem_amd.vae_model import HipTokenizer
    vae = _vae(384, 8192, 3, 224, seed=3)
    tok = HipTokenizer(vae, max_batch=4, precision="bf16")
    img = torch.rand(4, 3, 224, 224, device="cuda")
    ids = tok.get_codebook_indices(img)
    ref = vae.get_codebook_indices(img)
    assert ids.shape == (4, 196)
    assert (ids == ref).float().mean().item() >= 0.97


def test_tokenizer_bf16_mode_vs_reference_golden():
    """ids written by the reference DiscreteVAE (tests/golden/vae_tiny.npz, fp32): the opt-in bf16 mode must
    agree except where the reference's own top-2 logit gap is small."""
    import os
    import numpy as np
    from oracle.vae_ref import TINY_VAE, fill_vae_by_name, vae_inputs
    from mem_amd.vae_model import DiscreteVAE, HipTokenizer
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_tiny.npz"))
    m = DiscreteVAE(**TINY_VAE).eval()
    m.load_state_dict(fill_vae_by_name(m.state_dict(), seed=0))
    tok = HipTokenizer(m.cuda(), max_batch=6, precision="bf16")
    ids = tok.get_codebook_indices(vae_inputs(TINY_VAE, 6, 11).cuda()).cpu().numpy()
    want, gap = g["ids"], g["top2_gap"]
    diff = ids != want
    assert diff.mean() <= 0.03, diff.mean()
    spread = float(g["logits_b0"].std())
    assert (gap[diff] <= 0.05 * spread + 1e-3).all(), (gap[diff].max(), spread)


def test_tokenizer_fp16x2_mode():
    """Opt-in `fp16x2` mode (csrc/conv_f16x2.hip: two fp16 planes per value, three fp16 MFMAs per product): ids equal the
    REFERENCE's fp32 ids on both fixtures; logits within 4e-5 (at a logit spread of 1.77) of the fp32 HIP mode; layer-level:
    one convolution vs torch in float64 at 3e-6 of the output scale (fp32: 4e-7 sqrt(K))."""
    import os
    import numpy as np
    from mem_amd import ops
    from oracle.vae_ref import BASE_VAE, TINY_VAE, fill_vae_by_name, vae_inputs
    from mem_amd.vae_model import DiscreteVAE, HipTokenizer
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    for cfg, seed, fix, B in ((TINY_VAE, 0, "vae_tiny.npz", 6), (BASE_VAE, 1, "vae_base.npz", 2)):
        g = np.load(os.path.join(gdir, fix))
        m = DiscreteVAE(**cfg).eval()
        m.load_state_dict(fill_vae_by_name(m.state_dict(), seed=seed))
        m = m.cuda()
        img = vae_inputs(cfg, 6, 11) if cfg is TINY_VAE else vae_inputs(cfg, 2, 12) * (vae_inputs(cfg, 2, 13) < 0.3)
        tok = HipTokenizer(m, max_batch=B, precision="fp16x2")
        ids = tok.get_codebook_indices(img.cuda()).cpu().numpy()
        assert np.array_equal(ids, g["ids"]), fix
        lg2 = tok.logits.clone()
        tok32 = HipTokenizer(m, max_batch=B)
        tok32.get_codebook_indices(img.cuda())
        d = (lg2 - tok32.logits).abs().max().item()
        print(fix, "max |logit(fp16x2) - logit(fp32)| = %.2e, logit std %.3f" % (d, tok32.logits.std().item()))
        assert d <= 4e-5 * max(1.0, tok32.logits.std().item())
    # one layer against float64
    g = torch.Generator(device="cuda").manual_seed(5)
    Bn, cin, cout, h, k, s, p = 3, 128, 192, 14, 3, 1, 1
    x = torch.randn(Bn, cin, h, h, generator=g, device="cuda")
    w = torch.randn(cout, cin, k, k, generator=g, device="cuda") * 0.05
    b = torch.randn(cout, generator=g, device="cuda")
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p)

    def planes(t):
        hi = t.half()
        return torch.stack([hi, ((t - hi.float()) * 2048.0).half()]).contiguous()
    xp = torch.zeros(Bn, h + 2, h + 2, cin, device="cuda")
    xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    x2 = planes(xp)
    w2 = planes(w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous())
    dense = torch.zeros(Bn * h * h, cout, device="cuda")
    ops.conv2d_nhwc_f16x2(x2, w2, b, dense, Bn, h, h, cin, cout, k, s, p, relu=False, out_padded=False)
    err = (dense.view(Bn, h, h, cout).permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err <= 3e-6 * float(ref.abs().max()), err
    out2 = torch.zeros(2, Bn, h + 2, h + 2, cout, dtype=torch.float16, device="cuda")
    ops.conv2d_nhwc_f16x2(x2, w2, b, out2, Bn, h, h, cin, cout, k, s, p, relu=True)
    got = (out2[0].float() + out2[1].float() / 2048.0)[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double()
    assert (got - torch.relu(ref)).abs().max().item() <= 3e-6 * float(ref.abs().max())
    assert out2[:, :, 0].abs().max() == 0 and out2[:, :, :, 0].abs().max() == 0


def test_tok_flag_samples_list_is_ordered_and_complete():
    """csrc/conv_f32.hip::tok_flag_samples_kernel: the samples holding a token with gap <= kappa * rms (or a NaN), ascending, for
    batches below and above the kernel's group of 1024 samples; the statistics words accumulate."""
    from mem_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    stats = torch.zeros((4,), dtype=torch.int64, device="cuda")
    total = 0
    for B, hw in ((1, 196), (40, 196), (256, 196), (1500, 49), (2500, 7)):
        gap = torch.rand((B, hw), generator=g, device="cuda") + 0.01
        rms = torch.rand((B, hw), generator=g, device="cuda") + 0.5
        kappa = 0.05
        bad = torch.rand((B,), generator=g, device="cuda") < 0.03
        pos = torch.randint(0, hw, (B,), generator=g, device="cuda")
        gap[torch.arange(B, device="cuda")[bad], pos[bad]] = 0.0                # a planted near-tie
        if B > 8:
            gap[7, hw - 1] = float("nan")
        want = torch.nonzero(~(gap > kappa * rms).all(1)).view(-1).to(torch.int32)
        lst = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda")
        ops.tok_flag_samples(gap, rms, B, hw, kappa, lst, cnt, stats)
        n = int(cnt.item())
        assert n == want.numel(), (B, n, want.numel())
        assert torch.equal(lst[:n], want)
        total += n
    assert stats.tolist()[:2] == [total, 5]


def test_fp16x2_wide_tile_equals_the_128_tile_bit_for_bit():
    """csrc/conv_f16x2.hip::conv_gemm_f16x2_wide_kernel (256 x 128 x 32 tile, phase-interleaved waves) is chosen by grid size, which
    the small fixtures never reach: option conv_waves = 32 forces it.  Same k order per accumulator -> logits and ids bit-equal to
    the 128 x 128 kernel on both fixtures (ragged row tiles: 6 x 64, 2 x 196 output pixels; ResBlock add, 1 x 1 head), and one ragged
    layer (588 rows, 192 output channels = 1.5 column tiles) in all three epilogue forms."""
    import numpy as np
    from mem_amd import _lib, ops
    from oracle.vae_ref import BASE_VAE, TINY_VAE, fill_vae_by_name, vae_inputs
    from mem_amd.vae_model import DiscreteVAE, HipTokenizer
    try:
        for cfg, seed, B in ((TINY_VAE, 0, 6), (BASE_VAE, 1, 2)):
            m = DiscreteVAE(**cfg).eval()
            m.load_state_dict(fill_vae_by_name(m.state_dict(), seed=seed))
            m = m.cuda()
            img = (vae_inputs(cfg, 6, 11) if cfg is TINY_VAE else vae_inputs(cfg, 2, 12)).cuda()
            tok = HipTokenizer(m, max_batch=B, precision="fp16x2", certify=False)
            res = {}
            for w in (8, 32):
                _lib.set_option("conv_waves", w)
                ids = tok.get_codebook_indices(img).clone()
                res[w] = (ids, tok.logits.clone())
            assert torch.equal(res[8][0], res[32][0]) and torch.equal(res[8][1], res[32][1])
        g = torch.Generator(device="cuda").manual_seed(6)
        Bn, cin, cout, h, k, s, p = 3, 128, 192, 14, 3, 1, 1
        x = torch.randn(Bn, cin, h, h, generator=g, device="cuda")
        wt = torch.randn(cout, cin, k, k, generator=g, device="cuda") * 0.05
        b = torch.randn(cout, generator=g, device="cuda")

        def planes(t):
            hi = t.half()
            return torch.stack([hi, ((t - hi.float()) * 2048.0).half()]).contiguous()
        xp = torch.zeros(Bn, h + 2, h + 2, cin, device="cuda")
        xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
        x2, w2 = planes(xp), planes(wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous())
        addp = planes(torch.randn(Bn, h + 2, h + 2, cout, generator=g, device="cuda"))
        outs = {}
        for w in (8, 32):
            _lib.set_option("conv_waves", w)
            dense = torch.full((Bn * h * h, cout), 7.0, device="cuda")
            ops.conv2d_nhwc_f16x2(x2, w2, b, dense, Bn, h, h, cin, cout, k, s, p, relu=False, out_padded=False)
            o_relu = torch.zeros(2, Bn, h + 2, h + 2, cout, dtype=torch.float16, device="cuda")
            ops.conv2d_nhwc_f16x2(x2, w2, b, o_relu, Bn, h, h, cin, cout, k, s, p, relu=True)
            o_add = torch.zeros(2, Bn, h + 2, h + 2, cout, dtype=torch.float16, device="cuda")
            ops.conv2d_nhwc_f16x2(x2, w2, b, o_add, Bn, h, h, cin, cout, k, s, p, relu=False, add2=addp)
            outs[w] = (dense, o_relu, o_add)
        for a, c in zip(outs[8], outs[32]):
            assert torch.equal(a, c)
        assert outs[32][1][:, :, 0].abs().max() == 0 and outs[32][1][:, :, :, 0].abs().max() == 0     # the zero border stays zero
    finally:
        _lib.set_option("conv_waves", 16)


def test_tokenizer_fp16x2_labels_equal_fp32_on_rasterised_streams():
    """The entrypoint's default tokenizer mode (--tokenizer_impl hip_fp16x2) against the fp32-operand mode on the ViT-B
    tokenizer shape (hidden 384, 3 ResBlocks, 8192 tokens, 224^2) over 2 x 256 rasterised synthetic event streams (positive /
    time-surface / negative planes; the second batch mirrored in x) = 100 352 tokens: every label equal (bench.py reports the
    same statistic as with_tokenizer.label_mismatch_per_million_vs_fp32_mode), although near-ties exist in the set (smallest
    top-2 gap of the fp32 logits below 1e-6)."""
    import numpy as np
    from mem_amd import datasets as D
    from mem_amd.vae_model import DiscreteVAE, HipTokenizer
    torch.manual_seed(3)
    B, NE, H, W = 256, 30000, 224, 224
    vae = DiscreteVAE(input_H=H, input_W=W, num_tokens=8192, codebook_dim=512, num_layers=4, num_resnet_blocks=3,
                      hidden_dim=384, channels=3).cuda().eval()
    g = np.random.default_rng(1234)
    ev = np.empty((B * NE, 4), dtype=np.float64)
    ev[:, 0] = g.integers(0, W, B * NE)
    ev[:, 1] = g.integers(0, H, B * NE)
    ev[:, 2] = np.sort(g.integers(0, 300000, (B, NE)), axis=1).reshape(-1)
    ev[:, 3] = g.integers(0, 2, B * NE) * 2 - 1
    ev = torch.from_numpy(ev).cuda()
    off = (torch.arange(B + 1, dtype=torch.int64) * NE).cuda()
    imgs = [D.rasterize(ev, off, H, W, True, strict=False).float() / 255.0]
    ev[:, 0] = (W - 1) - ev[:, 0]
    imgs.append(D.rasterize(ev, off, H, W, True, strict=False).float() / 255.0)
    t32 = HipTokenizer(vae, max_batch=B)
    ids32 = [t32.get_codebook_indices(im).clone() for im in imgs]
    gap = min(float(t32.last_top2_gap(B).min()) for _ in (0,))
    del t32
    t16 = HipTokenizer(vae, max_batch=B, precision="fp16x2")
    bad = sum(int((t16.get_codebook_indices(im) != a).sum()) for im, a in zip(imgs, ids32))
    n = sum(a.numel() for a in ids32)
    assert n >= 100_000
    assert bad == 0, f"{bad} of {n} labels differ (smallest fp32 top-2 gap of the last batch {gap:.2e})"


def test_certified_tokenizer_planted_near_ties():
    """Round 5: the fp16x2 mode is CERTIFIED -- a label is kept only where its top-2 gap exceeds CERT_KAPPA x the row rms, every
    sample with a token below that margin is recomputed on the fp32 path on the device.  Near-ties are PLANTED: the upper half
    of the 512 classes are copies of the lower half with a relative weight perturbation of 1e-5 (independent rounding on both
    sides of every pair, twin gaps from ~1e-8 to ~1e-4 of the logit scale), so nearly every token sits at a near-tie between a
    class and its twin.  The certified ids must equal the fp32 mode's on EVERY token; the capacity of the fp32 recompute (8
    samples per round) is far below the number of flagged samples (every one of 40), so the multi-round path is what runs."""
    from mem_amd.vae_model import HipTokenizer
    vae = _vae(64, 512, 2, 64, seed=4)
    head = vae.encoder[-1]
    g = torch.Generator(device="cuda").manual_seed(9)
    with torch.no_grad():
        head.weight[256:] = head.weight[:256] * (1.0 + 1e-5 * torch.randn(head.weight[:256].shape, generator=g, device="cuda"))
        head.bias[256:] = head.bias[:256]
    B = 40
    img = torch.rand(B, 3, 64, 64, generator=g, device="cuda")
    t32 = HipTokenizer(vae, max_batch=B)
    ids32 = t32.get_codebook_indices(img).clone()
    gap32 = t32.last_top2_gap(B)
    rms = t32.logits.float().pow(2).mean(1).sqrt().view(B, -1)
    rel = (gap32 / rms).flatten()
    assert (rel < 7e-5).float().mean() > 0.8 and rel.min() < 1e-6 and rel.max() > 1e-7, (rel.min(), rel.median(), rel.max())
    raw = HipTokenizer(vae, max_batch=B, precision="fp16x2", certify=False)
    n_raw = int((raw.get_codebook_indices(img) != ids32).sum())
    cert = HipTokenizer(vae, max_batch=B, precision="fp16x2", exact_capacity=8)
    assert cert._exact.max_batch == 8
    ids = cert.get_codebook_indices(img)
    st = cert.certification_stats()
    print(f"planted near-ties: raw fp16x2 differs from fp32 on {n_raw} of {ids32.numel()} labels; flagged samples {st}")
    assert torch.equal(ids, ids32)
    assert st["flagged_samples"] == B and st["calls"] == 1
    # a second, smaller batch through the same object (stale slots of the earlier rounds must not leak), and a batch of one
    ids_b = cert.get_codebook_indices(img[5:18])
    assert torch.equal(ids_b, ids32[5:18])
    assert torch.equal(cert.get_codebook_indices(img[39:40]), ids32[39:40])


def test_certified_tokenizer_margin_and_unflagged_path():
    """(a) The stated error model: max |logit_fp16x2 - logit_fp32| relative to the row rms, measured on both reference
    fixtures and on random weights, stays below CERT_KAPPA / 4 (the margin is 2 x a 2 x padded bound).  (b) Samples that are
    NOT flagged keep their fp16x2 labels and these equal the fp32 ids (the reference fixtures: no token near a tie), with zero
    samples recomputed; ids == the reference's golden ids."""
    import os
    import numpy as np
    from oracle.vae_ref import BASE_VAE, TINY_VAE, fill_vae_by_name, vae_inputs
    from mem_amd.vae_model import DiscreteVAE, HipTokenizer
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    worst = 0.0
    for cfg, seed, fix, B in ((TINY_VAE, 0, "vae_tiny.npz", 6), (BASE_VAE, 1, "vae_base.npz", 2)):
        g = np.load(os.path.join(gdir, fix))
        m = DiscreteVAE(**cfg).eval()
        m.load_state_dict(fill_vae_by_name(m.state_dict(), seed=seed))
        m = m.cuda()
        img = (vae_inputs(cfg, 6, 11) if cfg is TINY_VAE else vae_inputs(cfg, 2, 12) * (vae_inputs(cfg, 2, 13) < 0.3)).cuda()
        tok = HipTokenizer(m, max_batch=B, precision="fp16x2")
        ids = tok.get_codebook_indices(img).cpu().numpy()
        assert np.array_equal(ids, g["ids"]), fix
        st = tok.certification_stats()
        tok32 = HipTokenizer(m, max_batch=B)
        tok32.get_codebook_indices(img)
        rms = tok32.logits.pow(2).mean(1, keepdim=True).sqrt()
        dev = ((tok.logits - tok32.logits).abs() / rms).max().item()
        relgap = (tok32.last_top2_gap(B).flatten() / rms.flatten()).min().item()
        print(f"{fix}: max deviation / rms = {dev:.2e}, smallest gap / rms = {relgap:.2e}, stats {st}")
        worst = max(worst, dev)
        assert tok.kappa >= 4.0 * tok.calibration["max_deviation_over_rms"] and 2.0 * dev <= tok.kappa, (tok.kappa, dev)
        if relgap > 2 * tok.kappa:
            assert st["flagged_samples"] == 0, st
    for hidden, tokens, res, size in ((64, 512, 2, 64), (128, 1024, 1, 96)):
        vae = _vae(hidden, tokens, res, size, seed=2)
        img = torch.rand(4, 3, size, size, device="cuda")
        a = HipTokenizer(vae, max_batch=4, precision="fp16x2", certify=False)
        b = HipTokenizer(vae, max_batch=4)
        a.get_codebook_indices(img); b.get_codebook_indices(img)
        rms = b.logits.pow(2).mean(1, keepdim=True).sqrt()
        worst = max(worst, ((a.logits - b.logits).abs() / rms).max().item())
    print(f"worst deviation / rms {worst:.2e} vs CERT_KAPPA {HipTokenizer.CERT_KAPPA:.1e}")
    assert worst * 4 <= HipTokenizer.CERT_KAPPA, worst


def test_certified_tokenizer_kappa_is_calibrated_per_model_and_audited():
    """Round 6: the certification margin is MEASURED PER MODEL at construction (4 x the worst fp16x2-vs-fp32 logit deviation on a
    calibration batch, floor 1e-5) instead of one constant for every tokenizer, and the claim is audited at run time.  On the
    stock random-weight model and on one whose first convolution is rescaled by 100 (another dynamic range in every layer
    behind it): the deviation measured on a DIFFERENT batch stays below kappa / 2 (the condition for equal labels), the
    certified ids equal the fp32 ids on every token, and the audit (every call here) recomputes samples without a mismatch."""
    from mem_amd.vae_model import HipTokenizer
    for scale0 in (1.0, 100.0):
        vae = _vae(64, 512, 2, 64, seed=6)
        with torch.no_grad():
            vae.encoder[0][0].weight.mul_(scale0)
            vae.encoder[0][0].bias.mul_(scale0)
        g = torch.Generator(device="cuda").manual_seed(31)
        img = torch.rand(16, 3, 64, 64, generator=g, device="cuda")
        img[8:] *= (torch.rand(8, 3, 64, 64, generator=g, device="cuda") < 0.3)
        t32 = HipTokenizer(vae, max_batch=16)
        ids32 = t32.get_codebook_indices(img).clone()
        rms = t32.logits.pow(2).mean(1, keepdim=True).sqrt()
        raw = HipTokenizer(vae, max_batch=16, precision="fp16x2", certify=False)
        raw.get_codebook_indices(img)
        dev = ((raw.logits - t32.logits).abs() / rms).max().item()
        tok = HipTokenizer(vae, max_batch=16, precision="fp16x2", audit_every=1)
        assert tok.kappa >= HipTokenizer.KAPPA_FLOOR and tok.calibration["samples"] == 8
        print(f"first conv x{scale0:g}: calibrated kappa {tok.kappa:.2e} (calibration deviation {tok.calibration['max_deviation_over_rms']:.2e}), "
              f"deviation on the test batch {dev:.2e}, logit rms {rms.mean().item():.3g}")
        assert 2.0 * dev <= tok.kappa, (dev, tok.kappa)
        for _ in range(3):
            assert torch.equal(tok.get_codebook_indices(img), ids32)
        st = tok.certification_stats()
        assert st["calls"] == 3 and st["audited_samples"] == 3 and st["audit_mismatches"] == 0, st
    # an explicit kappa switches the calibration off (the round-5 behaviour)
    t = HipTokenizer(vae, max_batch=4, precision="fp16x2", kappa=HipTokenizer.CERT_KAPPA)
    assert t.kappa == HipTokenizer.CERT_KAPPA and not hasattr(t, "calibration")
