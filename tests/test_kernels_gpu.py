"""-m gpu: every non-GEMM ViT kernel against a plain torch fp32 reference of the same op."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cuda") * scale


@pytest.mark.parametrize("R,D", [(394, 768), (50, 128), (197 * 3 + 1, 1024)])
def test_layernorm_fwd_bwd(R, D):
    from mem_amd import ops
    x = _rand((R + 5, D), 1, 2.0) + 0.5
    gamma, beta = 1 + _rand((D,), 2, 0.1), _rand((D,), 3, 0.1)
    y = torch.zeros((R, D), dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.zeros(R, device="cuda"), torch.zeros(R, device="cuda")
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, R, D)
    xr = x[:R].clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), gr, br, 1e-6)
    torch.testing.assert_close(y.float(), ref.bfloat16().float(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(mean, x[:R].mean(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rstd, 1 / torch.sqrt(x[:R].var(1, unbiased=False) + 1e-6), rtol=1e-4, atol=1e-5)
    dy = _rand((R, D), 4).bfloat16()
    ref.backward(dy.float())
    dres0 = _rand((R + 5, D), 5)
    dres = dres0.clone()
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dg, db, R, D, accumulate=True)
    torch.testing.assert_close(dres[:R] - dres0[:R], xr.grad, rtol=1e-3, atol=2e-4)
    assert torch.equal(dres[R:], dres0[R:])
    torch.testing.assert_close(dg, gr.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(db, br.grad, rtol=2e-3, atol=2e-3)


def test_layernorm_gathered_rows():
    from mem_amd import ops
    Rin, D = 400, 256
    x = _rand((Rin, D), 6)
    gamma, beta = 1 + _rand((D,), 7, 0.1), _rand((D,), 8, 0.1)
    idx = torch.tensor([3, 399, 17, 18, 250, 1], dtype=torch.int32, device="cuda")
    R = idx.numel()
    y = torch.zeros((R, D), dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.zeros(R, device="cuda"), torch.zeros(R, device="cuda")
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd, R, D, row_idx=idx)
    ref = F.layer_norm(x[idx.long()], (D,), gamma, beta, 1e-6)
    torch.testing.assert_close(y.float(), ref.bfloat16().float(), rtol=1e-2, atol=1e-2)
    dy = _rand((R, D), 9).bfloat16()
    dres = torch.zeros((Rin, D), device="cuda")
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dg, db, R, D, accumulate=False, row_idx=idx)
    xr = x.clone().requires_grad_(True)
    F.layer_norm(xr[idx.long()], (D,), gamma, beta, 1e-6).backward(dy.float())
    torch.testing.assert_close(dres, xr.grad, rtol=1e-3, atol=2e-4)


def test_branch_bwd():
    from mem_amd import ops
    T, Bn, D = 17, 7, 256
    M = T * Bn
    dx, y, gamma = _rand((M, D), 10), _rand((M, D), 11).bfloat16(), _rand((D,), 12, 0.2)
    keep = (torch.arange(Bn, device="cuda") % 2 == 0).float()
    for mask, kp in ((None, 1.0), (keep, 0.8)):
        dy = torch.zeros((M, D), dtype=torch.bfloat16, device="cuda")
        dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        ops.branch_bwd(dx, y, gamma, dy, dg, db, M, D, rowmask=mask, keep_prob=kp, rows_per_sample=T)
        dt = dx if mask is None else dx * mask.repeat_interleave(T).view(-1, 1) / kp
        torch.testing.assert_close(dy.float(), (dt * gamma).bfloat16().float(), rtol=1e-2, atol=1e-3)
        torch.testing.assert_close(dg, (dt * y.float()).sum(0), rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(db, dy.float().sum(0), rtol=1e-3, atol=1e-3)


def test_embed_bwd_and_fill_cls_and_im2col():
    from mem_amd import ops
    Bn, L, D = 4, 16, 128
    dx = _rand((Bn * (L + 1), D), 13)
    mask = (torch.rand((Bn * L,), device="cuda") < 0.5).to(torch.uint8)
    dy = torch.zeros((Bn * L, D), dtype=torch.bfloat16, device="cuda")
    dcls, dmt = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.embed_bwd(dx, mask, Bn, L, D, dy, dcls, dmt)
    d3 = dx.view(Bn, L + 1, D)
    w = mask.float().view(Bn, L, 1)
    torch.testing.assert_close(dcls, d3[:, 0].sum(0), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dmt, (d3[:, 1:] * w).sum((0, 1)), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dy.float().view(Bn, L, D), (d3[:, 1:] * (1 - w)).bfloat16().float(), rtol=0, atol=0)
    x = torch.zeros((Bn * (L + 1), D), device="cuda")
    cls = _rand((D,), 14)
    ops.fill_cls(x, Bn, L + 1, D, cls)
    assert torch.equal(x.view(Bn, L + 1, D)[:, 0], cls.expand(Bn, D)) and x.view(Bn, L + 1, D)[:, 1:].abs().sum() == 0
    img = _rand((3, 2, 64, 48), 15)
    out = torch.zeros((3 * 4 * 3, 2 * 256), dtype=torch.bfloat16, device="cuda")
    ops.im2col(img, 3, 2, 64, 48, 16, 16, out)
    ref = F.unfold(img, kernel_size=16, stride=16).transpose(1, 2).reshape(3 * 12, 512)
    torch.testing.assert_close(out.float(), ref.bfloat16().float(), rtol=0, atol=0)


@pytest.mark.parametrize("M,V", [(600, 8192), (37, 512), (5, 1024)])
def test_cross_entropy(M, V):
    from mem_amd import ops
    logits = (_rand((M, V), 16, 2.0)).bfloat16()
    labels = torch.randint(0, V, (M,), device="cuda")
    logits[0, labels[0]] = 30.0            # a surely-correct row
    lg = logits.clone()
    row_loss, row_ok = torch.zeros(M, device="cuda"), torch.zeros(M, dtype=torch.int32, device="cuda")
    out2 = torch.zeros(2, device="cuda")
    ops.cross_entropy(lg, labels, M, V, 1.0 / M, row_loss, row_ok, out2)
    lr_ = logits.float().requires_grad_(True)
    loss = F.cross_entropy(lr_, labels)
    loss.backward()
    torch.testing.assert_close(out2[0], loss.detach(), rtol=1e-5, atol=1e-5)
    acc = (logits.float().cpu().max(-1)[1] == labels.cpu()).float().mean()
    assert abs(out2[1].item() - acc.item()) < 1e-6
    torch.testing.assert_close(lg.float(), lr_.grad.bfloat16().float(), rtol=2e-2, atol=1e-7)


def test_cross_entropy_label_out_of_range_is_nan_not_a_fault():
    """torch raises "Target out of bounds"; the device kernel must not read logits[label] then: the loss becomes NaN,
    which the training loop's non-finite-loss abort reports at its next meter flush."""
    from mem_amd import ops
    M, V = 64, 512
    logits = _rand((M, V), 17, 2.0).bfloat16()
    labels = torch.randint(0, V, (M,), device="cuda")
    labels[7] = 2147483647
    labels[9] = -5
    row_loss, row_ok = torch.zeros(M, device="cuda"), torch.zeros(M, dtype=torch.int32, device="cuda")
    out2 = torch.zeros(2, device="cuda")
    ops.cross_entropy(logits, labels, M, V, 1.0 / M, row_loss, row_ok, out2)
    torch.cuda.synchronize()
    assert torch.isnan(row_loss[7]) and torch.isnan(row_loss[9]) and torch.isnan(out2[0])
    good = torch.ones(M, dtype=torch.bool, device="cuda"); good[7] = good[9] = False
    assert torch.isfinite(row_loss[good]).all() and torch.isfinite(logits.float()).all()


def _attn_ref(qkv, bias, B, T, D, H, scale):
    """reference attention in fp32 with the autocast rounding points (bf16 matmul outputs)."""
    q, k, v = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)).bfloat16().float() + bias
    p = s.softmax(-1)
    o = (p.bfloat16().float() @ v).bfloat16()
    return o.transpose(1, 2).reshape(B * T, D), p


@pytest.mark.parametrize("B,T,H,win", [(3, 197, 12, (14, 14)), (45, 197, 4, (14, 14)), (2, 17, 2, (4, 4)), (5, 65, 4, (8, 8)),
                                       (1, 256, 2, (15, 17)), (40, 37, 3, (4, 9)),
                                       # > 256 tokens: the streaming kernels (attn_stream.hip); 30x40 = ViT-L @ 480x640
                                       (1, 257, 2, (16, 16)), (3, 324, 3, (17, 19)), (2, 1201, 2, (30, 40)),
                                       (64, 290, 16, (17, 17)),
                                       # windows 40 / 20 wide: the slot-layout kernels (attn_win.hip, round 5); ragged last chunks
                                       # (7 = 2 x 3 + 1 grid rows, 13 = 2 x 5 + 3), workgroups persistent over several samples
                                       (2, 321, 2, (16, 20)), (3, 261, 3, (13, 20)), (4, 281, 3, (7, 40)), (40, 1201, 16, (30, 40)),
                                       # (round 6) the ragged windows whose padded-row buckets reach the pad behind the table (the NaN
                                       # bug of round 5 lived there) against the ORACLE, not only against the stream kernels
                                       (9, 1041, 4, (26, 40)), (9, 921, 4, (23, 40)), (5, 281, 3, (7, 40))])
def test_attention_fwd_bwd(B, T, H, win):
    _attention_case(B, T, H, win)


@pytest.mark.parametrize("B,T,H,win", [(2, 321, 2, (16, 20)), (3, 261, 3, (13, 20)), (4, 281, 3, (7, 40)), (9, 1041, 4, (26, 40)),
                                       (9, 921, 4, (23, 40)), (3, 1201, 2, (30, 40)), (40, 1201, 16, (30, 40))])
def test_attention_bwd_ds_storing_form(B, T, H, win):
    """The dS-storing backward of the long-window kernels (memhip_attn_bwd_ws: the dK / dV kernel stores dS and owns the table
    gradient, the dQ kernel is a streaming product over it) against the same oracle as the recomputing form; B = 40 x 16 heads:
    workgroups persistent over several samples, more than 16 samples per workgroup split."""
    from mem_amd import ops
    nbytes = ops.attn_bwd_workspace(B, T, H, win)
    assert nbytes == B * H * ops.attn_tokens_padded(T) * 128 * -(-win[0] // (128 // ((win[1] + 7) // 8 * 8))) * 2
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ws.fill_(0xFF)                                  # NaN patterns: every word the dQ kernel reads must have been written
    _attention_case(B, T, H, win, ws=ws)


def test_attention_bwd_workspace_is_optional():
    """No workspace form for the short windows (0 bytes); a workspace that is too small falls back to the recomputing kernels."""
    from mem_amd import ops
    assert ops.attn_bwd_workspace(4, 197, 12, (14, 14)) == 0 and ops.attn_bwd_workspace(4, 325, 4, (18, 18)) == 0
    _attention_case(2, 321, 2, (16, 20), ws=torch.empty(1024, dtype=torch.uint8, device="cuda"))


def test_attn_win_equals_stream_kernels():
    """The slot-layout kernels (attn_win.hip) against the token-order streaming kernels (attn_stream.hip, option attn_win = 0)
    on the same inputs: equal to the rounding of the bf16 outputs (the two differ in fp32 summation order and in the
    truncating v_dot2c bias add), table gradient to its fixed-point step."""
    from mem_amd import _lib, ops
    from oracle.vit_ref import rel_pos_index
    # (26 and 23 rows of 40: ragged last chunks whose buckets reach the pad behind the table -- NaNs in dK / dV before the pad was zeroed)
    for B, H, win in ((3, 2, (30, 40)), (5, 3, (7, 40)), (4, 3, (13, 20)), (9, 4, (26, 40)), (9, 4, (23, 40)), (24, 16, (30, 40))):
        T, D = win[0] * win[1] + 1, 64 * H
        TP = ops.attn_tokens_padded(T)
        g = torch.Generator(device="cuda").manual_seed(B)
        qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.7)
        qkv[:, :D] *= 0.5
        qkv = qkv.bfloat16()
        _, nrd = rel_pos_index(win)
        table = torch.randn(nrd, H, generator=g, device="cuda") * 0.5
        dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
        res = {}
        try:
            for mode in (0, 1):
                _lib.set_option("attn_win", mode)
                out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
                dqkv = torch.full((B * T, 3 * D), 3.0, dtype=torch.bfloat16, device="cuda")
                dtable = torch.zeros(nrd, H, device="cuda"); dqb = torch.zeros(D, device="cuda"); dvb = torch.zeros(D, device="cuda")
                delta = torch.zeros(2 * B * T + 4, H, device="cuda")
                ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
                ops.attn_delta(dout, out, B * T, H, delta)
                ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dqb, dvb)
                torch.cuda.synchronize()
                res[mode] = (out.float(), lse[:, :, :T].clone(), dqkv.float(), dtable.clone(), dqb.clone(), dvb.clone())
        finally:
            _lib.set_option("attn_win", 1)
        for name, a, b, tol in zip(("out", "lse", "dqkv", "dtable", "dq_bias", "dv_bias"), res[0], res[1],
                                   (2e-3, 1e-6, 3e-3, 2e-3, 4e-3, 4e-3)):
            rel = ((a - b).norm() / (a.norm() + 1e-30)).item()
            assert torch.isfinite(b).all() and rel <= tol, (B, H, win, name, rel)


def test_attention_general_kernels_at_14x14():
    """The 14 x 14 window dispatches to attn16.hip; with the switch off the general kernels of attn.hip take it."""
    from mem_amd import _lib
    _lib.set_option("attn16", 0)
    try:
        _attention_case(29, 197, 3, (14, 14))
    finally:
        _lib.set_option("attn16", 1)


def test_attn16_equals_general_kernels():
    """The 14 x 14 kernels (key-slot layout, fused backward) against the general ones on the same inputs: the same arithmetic
    up to the order of fp32 sums and the fixed-point step of the table gradient."""
    from mem_amd import ops, _lib
    from oracle.vit_ref import rel_pos_index
    B, T, H, win = 29, 197, 3, (14, 14)
    D = 64 * H
    TP = ops.attn_tokens_padded(T)
    qkv = _rand((B * T, 3 * D), 60, 0.7)
    qkv[:, :D] *= 0.5
    qkv = qkv.bfloat16()
    idx, nrd = rel_pos_index(win)
    table = _rand((nrd, H), 61, 0.5)
    dout = _rand((B * T, D), 62, 1.0).bfloat16()
    res = {}
    try:
        for mode in (0, 1):
            _lib.set_option("attn16", mode)
            out = torch.zeros((B * T, D), dtype=torch.bfloat16, device="cuda")
            lse = torch.zeros((B, H, TP), device="cuda")
            dqkv = torch.full((B * T, 3 * D), 3.0, dtype=torch.bfloat16, device="cuda")
            dtable = torch.zeros((nrd, H), device="cuda")
            delta = torch.zeros((2 * B * T + 4, H), device="cuda")
            dqb = torch.zeros(D, device="cuda")
            ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
            ops.attn_delta(dout, out, B * T, H, delta)
            ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dq_bias=dqb)
            res[mode] = (out.float(), lse[:, :, :T].clone(), dqkv.float(), dtable.clone(), dqb.clone())
    finally:
        _lib.set_option("attn16", 1)
    for name, a, b, tol in zip(("out", "lse", "dqkv", "dtable", "dq_bias"), res[0], res[1], (1e-3, 1e-6, 1e-3, 5e-3, 2e-3)):
        rel = ((a - b).norm() / a.norm()).item()
        assert rel < tol, (name, rel)


@pytest.mark.parametrize("B,T,H,win,with_table", [(29, 197, 3, (14, 14), True), (14, 197, 12, (14, 14), False), (1, 197, 2, (14, 14), True),
                                                 (5, 25, 2, (4, 6), True), (2, 321, 2, (16, 20), True)])
def test_attention_backward_from_the_forward_output(B, T, H, win, with_table):
    """memhip_attn_bwd_out: rowsum(dO * O) computed by the library (inside the fused 14 x 14 backward; a delta pass in front
    of the general / streaming kernels) == memhip_attn_delta + memhip_attn_bwd on the same inputs (Attention.forward backward,
    mem/modeling_finetune.py:137-154).  B = 1 / 14 / 29: one, uneven and several samples per workgroup (first-sample path,
    the next-sample loads of the last sample)."""
    from mem_amd import ops
    from oracle.vit_ref import rel_pos_index
    D = 64 * H
    TP = ops.attn_tokens_padded(T)
    qkv = _rand((B * T, 3 * D), 160, 0.7)
    qkv[:, :D] *= 0.5
    qkv = qkv.bfloat16()
    idx, nrd = rel_pos_index(win)
    table = _rand((nrd, H), 161, 0.5)
    dout = _rand((B * T, D), 162, 1.0).bfloat16()
    out = torch.zeros((B * T, D), dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros((B, H, TP), device="cuda")
    ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
    res = []
    for fused in (False, True):
        dqkv = torch.full((B * T, 3 * D), 3.0, dtype=torch.bfloat16, device="cuda")
        dtable = torch.zeros((nrd, H), device="cuda") if with_table else None
        delta = torch.full((2 * B * T + 4, H), 7.0, device="cuda")             # garbage: the fused path must not read it
        dqb = torch.zeros(D, device="cuda")
        if fused:
            ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dq_bias=dqb, out=out)
        else:
            ops.attn_delta(dout, out, B * T, H, delta)
            ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dq_bias=dqb)
        res.append((dqkv.float(), dtable.clone() if with_table else torch.zeros(1, device="cuda"), dqb.clone()))
    for name, a, b, tol in zip(("dqkv", "dtable", "dq_bias"), res[0], res[1], (2e-4, 2e-4, 2e-4)):
        rel = ((a - b).norm() / (a.norm() + 1e-20)).item()
        assert rel < tol, (name, rel)
    assert (res[0][0] == res[1][0]).float().mean().item() > 0.999           # bf16 outputs: equal except rounding-boundary cases


def _attention_case(B, T, H, win, ws=None):
    from mem_amd import ops
    from oracle.vit_ref import rel_pos_index
    D = 64 * H
    scale = 0.125
    TP = ops.attn_tokens_padded(T)
    qkv = _rand((B * T, 3 * D), 20, 1.0)
    qkv[:, :D] *= scale
    qkv = qkv.bfloat16()
    idx, nrd = rel_pos_index(win)
    idx = idx.cuda()
    table = _rand((nrd, H), 21, 0.5)
    bias_pad = torch.zeros((H, TP, TP), device="cuda")
    biasT_pad = torch.zeros_like(bias_pad)
    ops.relpos_gather(table, idx.int().contiguous(), T, TP, H, bias_pad, biasT_pad)
    bias = table[idx.view(-1)].view(T, T, H).permute(2, 0, 1).contiguous()
    assert torch.equal(bias_pad[:, :T, :T], bias) and bias_pad[:, T:].abs().sum() == 0
    assert torch.equal(biasT_pad, bias_pad.transpose(1, 2).contiguous())
    out = torch.zeros((B * T, D), dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros((B, H, TP), device="cuda")
    ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)          # bias gathered on chip from the table
    ref, p = _attn_ref(qkv, bias, B, T, D, H, scale)
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-2, atol=2e-2)
    q, k, _ = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)).bfloat16().float() + bias
    torch.testing.assert_close(lse[:, :, :T], torch.logsumexp(s, -1), rtol=1e-4, atol=1e-4)
    # ---- backward against autograd through the same rounded-op reference
    dout = _rand((B * T, D), 22, 1.0).bfloat16()
    qf = qkv.float().requires_grad_(True)
    tb = table.clone().requires_grad_(True)
    bb = tb[idx.view(-1)].view(T, T, H).permute(2, 0, 1)
    q, k, v = qf.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    o = ((q @ k.transpose(-2, -1)) + bb).softmax(-1) @ v
    o.transpose(1, 2).reshape(B * T, D).backward(dout.float())
    dqkv = torch.zeros((B * T, 3 * D), dtype=torch.bfloat16, device="cuda")
    dtable = torch.zeros((nrd, H), device="cuda")
    delta = torch.zeros((2 * B * T + 4, H), device="cuda")  # delta, |dO|^2, 4 rows of per-head bounds
    ops.attn_delta(dout, out, B * T, H, delta)
    torch.testing.assert_close(delta[:B * T], (dout.float() * out.float()).view(B * T, H, 64).sum(-1), rtol=1e-4, atol=1e-3)
    dqb, dvb = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    # 14 x 14 window, B > 3: the engine's call (no v_bias gradient) = the fused backward of attn16.hip; with dv_bias the
    # general two-kernel backward of attn.hip runs (both forms are covered at 14 x 14)
    fused = win == (14, 14) and B > 3
    ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, scale, dqkv, dtable, dq_bias=dqb,
                 dv_bias=None if fused else dvb, ws=ws)
    torch.testing.assert_close(dqb, dqkv[:, :D].float().sum(0), rtol=1e-3, atol=1e-2)
    if not fused:
        torch.testing.assert_close(dvb, dqkv[:, 2 * D:].float().sum(0), rtol=1e-3, atol=1e-2)
    g = qf.grad.clone()
    g[:, :D] *= scale          # kernel returns d(q_lin) = d(q') * scale
    err = (dqkv.float() - g).abs().max().item()
    assert err < 0.05 * g.abs().max().item() + 1e-3, err
    rel = (dqkv.float() - g).norm() / g.norm()
    assert rel < 2e-2, rel
    relb = (dtable - tb.grad).norm() / tb.grad.norm()
    assert relb < 2e-2, relb


def test_residual_rows_and_scatter_rows():
    """The row kernels of the last block's tail form (ViTEngine.tail_rows): out[i] = x[rows[i]] + drop_path(gamma * y[i]) with
    the RESIDUAL epilogue's arithmetic -- gamma * y rounded once, the quotient by keep_prob correctly rounded (== IEEE
    division), times the row's keep flag, one rounding for the add -- so torch's fp32 ops in that order give the SAME bits;
    scatter_rows is a row copy."""
    from mem_amd import ops
    M, R, D = 5000, 1234, 768
    x = _rand((M, D), 300)
    y = _rand((R, D), 301).bfloat16()
    gamma = _rand((D,), 302, 0.1)
    rows = torch.randperm(M, device="cuda")[:R].sort().values.int()
    keep = (torch.rand(R, device="cuda") > 0.3).float()
    for g, rk, kp in ((gamma, keep, 0.9), (None, None, 1.0), (gamma, None, 0.7), (None, keep, 0.8)):
        out = torch.full((R, D), 7.0, device="cuda")
        ops.residual_rows(x, rows, y, g, rk, kp, R, D, out)
        br = (y.float() if g is None else g.view(1, D) * y.float()).cpu()      # (the quotient on the CPU: IEEE division)
        if rk is not None:
            br = br.div(kp) * rk.cpu().view(R, 1)
        ref = x.index_select(0, rows.long()).cpu() + br
        assert torch.equal(out.cpu(), ref), (g is not None, rk is not None, kp, float((out.cpu() - ref).abs().max()))
    dst = torch.zeros((M, D), device="cuda")
    src = _rand((R, D), 303)
    ops.scatter_rows(src, rows, R, D, dst)
    ref = torch.zeros((M, D), device="cuda")
    ref.index_copy_(0, rows.long(), src)
    assert torch.equal(dst, ref)


def test_cast_and_transposes():
    from mem_amd import ops
    w = _rand((2304, 768), 30)
    flat = torch.zeros(2304 * 768, dtype=torch.bfloat16, device="cuda")
    ops.cast_f32_bf16(w, flat, w.numel())
    assert torch.equal(flat.view(2304, 768), w.bfloat16())
    wt = torch.zeros((768, 2304), dtype=torch.bfloat16, device="cuda")
    ops.transpose_cast(w, 2304, 768, wt)
    assert torch.equal(wt, w.bfloat16().t().contiguous())
    R, Cc = 394, 2304
    Rp = 448
    a = _rand((R, Cc), 31).bfloat16()
    at = torch.full((Cc, Rp), 9.0, dtype=torch.bfloat16, device="cuda")
    cs0, cs1 = torch.zeros(768, device="cuda"), torch.zeros(768, device="cuda")
    ops.transpose_bf16(a, R, Cc, at, Rp, cs0, (0, 768), cs1, (1536, 2304))
    assert torch.equal(at[:, :R], a.t().contiguous()) and at[:, R:].abs().sum() == 0
    torch.testing.assert_close(cs0, a[:, :768].float().sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(cs1, a[:, 1536:].float().sum(0), rtol=1e-4, atol=1e-3)


def test_grad_norm_and_adamw_match_torch():
    from mem_amd import ops
    n = 1024 * 37
    p0, g = _rand((n,), 40), _rand((n,), 41, 3.0)
    flags = torch.zeros(n // 1024, dtype=torch.uint8, device="cuda")
    flags[::2] = 1
    ws = torch.zeros(1024, dtype=torch.float64, device="cuda")
    norm = torch.zeros(1, device="cuda")
    ops.grad_norm(g, n, norm, ws)
    torch.testing.assert_close(norm[0], g.double().norm().float(), rtol=1e-6, atol=0)
    # torch reference on CPU (true-division semantics), two param groups
    pc = p0.cpu().clone()
    dec = flags.cpu().bool().repeat_interleave(1024)
    pa, pb = torch.nn.Parameter(pc[dec].clone()), torch.nn.Parameter(pc[~dec].clone())
    opt = torch.optim.AdamW([{"params": [pa], "weight_decay": 0.05}, {"params": [pb], "weight_decay": 0.0}],
                            lr=5e-4, betas=(0.9, 0.95), eps=1e-8)
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gs = g * step
        ops.grad_norm(gs, n, norm, ws)
        ops.adamw(p, gs, m, v, n, flags, 5e-4, 0.9, 0.95, 1e-8, 0.05, step, gnorm=norm, max_norm=30.0)
        gc = gs.cpu()
        pa.grad, pb.grad = gc[dec].clone(), gc[~dec].clone()
        torch.nn.utils.clip_grad_norm_([pa, pb], 30.0)
        opt.step()
    got = p.cpu()
    torch.testing.assert_close(got[dec], pa.detach(), rtol=2e-6, atol=2e-7)
    torch.testing.assert_close(got[~dec], pb.detach(), rtol=2e-6, atol=2e-7)


def test_attention_full_size_properties():
    """BASELINE size (B=256, 197 tokens, 12 heads): size-independent properties instead of a reference tensor.
    (1) softmax rows sum to one: with V == const the output is that constant; (2) lse is the log of the row sum of
    exp(score): recomputed on a few sampled (sample, head, query) rows; (3) backward with dO == 0 gives exactly zero
    gradients (and overwrites stale output); (4) softmax-backward rows sum to zero, so the bucket gradients of every
    head sum to zero up to fixed-point / bf16 noise (random V: with V == const the true gradient itself vanishes and
    only the rounding difference between bf16 dP and the flash-form delta would be left)."""
    from mem_amd import ops
    from oracle.vit_ref import rel_pos_index
    B, T, H, win = 256, 197, 12, (14, 14)
    D = 64 * H
    TP = ops.attn_tokens_padded(T)
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.5)
    qkv[:, 2 * D:] = 0.75                                        # V == const
    qkv = qkv.bfloat16()
    idx, nrd = rel_pos_index(win)
    table = torch.randn(nrd, H, generator=g, device="cuda") * 0.3
    out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.zeros(B, H, TP, device="cuda")
    ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
    assert (out.float() - 0.75).abs().max().item() <= 0.75 * 2 ** -7       # rows of P sum to 1 (bf16 P, fp32 accumulate)
    bias = table[idx.cuda().view(-1)].view(T, T, H).permute(2, 0, 1)
    for (b, h, q) in ((0, 0, 0), (17, 3, 101), (255, 11, 196), (128, 7, 1)):
        qv = qkv[b * T + q, h * 64:(h + 1) * 64].float()
        kv = qkv[b * T:(b + 1) * T, D + h * 64:D + (h + 1) * 64].float()
        s = (kv @ qv).bfloat16().float() + bias[h, q]
        assert abs(torch.logsumexp(s, 0).item() - lse[b, h, q].item()) <= 2e-3
    dqkv = torch.full((B * T, 3 * D), 7.0, dtype=torch.bfloat16, device="cuda")
    dtable = torch.zeros(nrd, H, device="cuda")
    delta = torch.zeros(2 * B * T + 4, H, device="cuda")
    dout = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda")
    ops.attn_delta(dout, out, B * T, H, delta)
    ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dq_bias=torch.zeros(D, device="cuda"))
    assert dqkv.float().abs().max().item() == 0.0 and dtable.abs().max().item() == 0.0
    qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.5).bfloat16()      # random V for the backward identity
    ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
    dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
    ops.attn_delta(dout, out, B * T, H, delta)
    ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dq_bias=torch.zeros(D, device="cuda"))
    # softmax-backward rows sum to zero, so each head's bucket gradients sum to (fixed-point / bf16 noise around) zero
    assert (dtable.sum(0).abs() <= 1e-2 * dtable.abs().sum(0) + 1e-3).all(), dtable.sum(0)
    assert torch.isfinite(dqkv.float()).all()


def test_attn16_random_shapes_and_repeated_launches():
    """tools/stress_attn16.py: random batch sizes / head counts / stagger settings against the general kernels, every run
    repeated for bitwise reproducibility, then the BASELINE-size launch 40 times with bitwise equal results (a stale LDS image
    behind a counted wait or a barrier race would not be reproducible)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_attn16.py"), "1", "16", "40"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "bitwise equal every time" in r.stdout


def test_attn_win_random_shapes_and_repeated_launches():
    """tools/stress_attn_win.py: random batch sizes / head counts / window heights at widths 40 and 20 against the token-order
    streaming kernels, every run repeated for bitwise reproducibility, then the config-#5 launch (64 x 16 heads x 1201 tokens,
    workgroups persistent over 4 samples) 6 times with bitwise equal forward / dQ / dK / dV."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_attn_win.py"), "1", "10", "6"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "bitwise equal every time" in r.stdout


def test_stream_reservation_changes_grids_not_results():
    """memhip_stream_reserve_cus (round 4): launches on a stream that carries a reservation size their persistent grids for fewer
    CUs -- the partition of tiles / samples over workgroups changes, the arithmetic does not: GEMM outputs and the attention
    forward are bit-equal, the attention backward equal up to the order of its fp32 atomics; another stream is unaffected and
    a reservation of 0 restores the full grid."""
    from mem_amd import ops
    from oracle.vit_ref import rel_pos_index
    M, N, K = 256 * 24, 768, 768
    A = _rand((M, K), 70).bfloat16(); W = _rand((N, K), 71, 0.05).bfloat16(); bias = _rand((N,), 72)
    B, T, H, win = 40, 197, 12, (14, 14)
    D = 64 * H
    TP = ops.attn_tokens_padded(T)
    qkv = _rand((B * T, 3 * D), 73, 0.6).bfloat16()
    idx, nrd = rel_pos_index(win)
    table = _rand((nrd, H), 74, 0.4)
    dout = _rand((B * T, D), 75).bfloat16()

    def run():
        o = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        ops.gemm_nt(A, W, M, N, K, ops.EPI_BIAS_BF16, out0=o, bias=bias)
        out = torch.zeros((B * T, D), dtype=torch.bfloat16, device="cuda")
        lse = torch.zeros((B, H, TP), device="cuda")
        dqkv = torch.zeros((B * T, 3 * D), dtype=torch.bfloat16, device="cuda")
        dtable = torch.zeros((nrd, H), device="cuda")
        delta = torch.zeros((2 * B * T + 4, H), device="cuda")
        dqb = torch.zeros(D, device="cuda")
        ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
        ops.attn_delta(dout, out, B * T, H, delta)
        ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dq_bias=dqb)
        torch.cuda.synchronize()
        return o, out, lse, dqkv, dtable
    base = run()
    st = torch.cuda.current_stream()
    try:
        ops.stream_reserve_cus(st, 32)
        res = run()
        other = torch.cuda.Stream()
        with torch.cuda.stream(other):
            oth = run()
    finally:
        ops.stream_reserve_cus(st, 0)
    again = run()
    for r in (res, oth, again):
        assert torch.equal(r[0], base[0]) and torch.equal(r[1], base[1]) and torch.equal(r[2], base[2])
        assert torch.equal(r[3], base[3])
        assert ((r[4] - base[4]).norm() / base[4].norm()).item() < 1e-4     # (fp32 summation order over samples / workgroups)
