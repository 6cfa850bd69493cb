"""Table gradient (and dqkv) of the window attention against torch autograd in float64 (no rounding points): which of the backward
forms (attn_win = 0 stream, 1 slot layout, 9 = 1 + dS workspace) is closer?  usage: attn_win_dtable_ref.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index

def run(B, H, win, seed):
    T, D = win[0] * win[1] + 1, 64 * H
    TP = ops.attn_tokens_padded(T)
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.7); qkv[:, :D] *= 0.5
    qkv = qkv.bfloat16()
    idx, nrd = rel_pos_index(win)
    table = torch.randn(nrd, H, generator=g, device="cuda") * 0.5
    dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
    # float64 autograd reference (q already carries the scale in this layout: the engine's qkv GEMM epilogue applies it)
    x = qkv.double().view(B, T, 3, H, 64).requires_grad_(True)
    tb = table.double().requires_grad_(True)
    q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
    bias = tb[torch.as_tensor(idx, device="cuda").view(-1)].view(T, T, H).permute(2, 0, 1)
    o = (torch.softmax(q @ k.transpose(-1, -2) + bias, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, D)
    o.backward(dout.double())
    dt_ref = tb.grad
    dx = x.grad.reshape(B * T, 3 * D).clone()
    for mode in (0, 1, 9):
        assert _lib.lib.memhip_set_option(b"attn_win", 1 if mode == 9 else mode) == 0
        out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
        dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
        delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
        ws = torch.empty(ops.attn_bwd_workspace(B, T, H, win), dtype=torch.uint8, device="cuda") if mode == 9 else None
        ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
        ops.attn_delta(dout, out, B * T, H, delta)
        ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 1.0, dqkv, dtable, dqb, None, ws=ws)
        torch.cuda.synchronize()
        e = (dtable.double() - dt_ref)
        cls = e[-3:].abs().max().item()
        print(f"B={B} H={H} win={win} mode {mode}: dtable rel-L2 vs f64 {(e.norm() / dt_ref.norm()).item():.3e} max {e.abs().max().item():.3e} "
              f"(cls buckets max err {cls:.3e});  dqkv rel-L2 {((dqkv.double() - dx).norm() / dx.norm()).item():.3e}", flush=True)
    assert _lib.lib.memhip_set_option(b"attn_win", 1) == 0

run(2, 4, (16, 20), 0)
run(3, 2, (30, 40), 1)
run(5, 3, (7, 40), 3)
run(4, 3, (13, 20), 4)
run(24, 16, (30, 40), 5)
