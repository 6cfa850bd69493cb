"""A/B of the slot-layout streaming attention (attn_win.hip, option attn_win = 1) against attn_stream.hip (attn_win = 0) on the
same inputs, forward also against a float64 torch evaluation of the reference formula; timing at the config-#5 size.
usage: attn_win_check.py [fwd|all] [time]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index

what = sys.argv[1] if len(sys.argv) > 1 else "all"
do_time = len(sys.argv) > 2
MODES = tuple(int(x) for x in os.environ.get('WIN_MODES', '0,1,2').split(','))

def setopt(v):
    assert _lib.lib.memhip_set_option(b"attn_win", 1 if v == 9 else v) == 0       # mode 9 = the kernels of mode 1 + a dS workspace

def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def run(B, H, win, seed=0, time_it=False, ref64=False):
    T, D = win[0] * win[1] + 1, 64 * H
    TP = ops.attn_tokens_padded(T)
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.7)
    qkv[:, :D] *= 0.5
    qkv = qkv.bfloat16()
    idx, nrd = rel_pos_index(win)
    table = torch.randn(nrd, H, generator=g, device="cuda") * 0.5
    dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
    res = {}
    for mode in MODES:
        setopt(mode)
        out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
        dqkv = torch.full((B * T, 3 * D), 3.0, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
        delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
        ws = torch.empty(ops.attn_bwd_workspace(B, T, H, win), dtype=torch.uint8, device="cuda") if mode == 9 else None
        ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)
        if what == "all":
            ops.attn_delta(dout, out, B * T, H, delta)
            ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dqb, None, ws=ws)
        torch.cuda.synchronize()
        res[mode] = (out.float(), lse[:, :, :T].clone(), dqkv.float(), dtable.clone(), dqb.clone())
        for rep in range(2 if time_it else 0):
            msg = f"mode {mode}: fwd {tm(lambda: ops.attn_fwd(qkv, B, T, D, H, table, win, out, lse)):.1f} us"
            if what == "all":
                msg += f"  bwd {tm(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, dtable, dqb, None, ws=ws)):.1f} us"
                msg += f"  bwd(no dtable) {tm(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, win, B, T, D, H, 0.125, dqkv, None, dqb, None, ws=ws)):.1f} us"
            print(msg, flush=True)
    names = ("out", "lse", "dqkv", "dtable", "dq_bias")
    for mode in MODES[1:]:
        for n, a, b in list(zip(names, res[MODES[0]], res[mode]))[: (5 if what == "all" else 2)]:
            d = (a - b).abs().max().item(); rel = ((a - b).norm() / (a.norm() + 1e-30)).item()
            print(f"B={B} H={H} win={win} mode {mode} {n}: max|old-new| {d:.3e}  rel-L2 {rel:.3e}  max|old| {a.abs().max().item():.3e}  finite {bool(torch.isfinite(b).all())}", flush=True)
    if ref64:
        q = qkv.view(B, T, 3, H, 64).double()
        qq, kk, vv = q[:, :, 0].permute(0, 2, 1, 3), q[:, :, 1].permute(0, 2, 1, 3), q[:, :, 2].permute(0, 2, 1, 3)
        bias = table.double()[torch.as_tensor(idx, device="cuda").view(-1)].view(T, T, H).permute(2, 0, 1)
        s = (qq @ kk.transpose(-1, -2)).bfloat16().double() + bias          # the matmul output is bf16 under autocast
        p = torch.softmax(s, -1)
        ref = (p.bfloat16().double() @ vv).permute(0, 2, 1, 3).reshape(B * T, D)
        lse_ref = torch.logsumexp(s, -1)
        for mode in MODES:
            a = res[mode][0].double()
            print(f"   vs float64 formula, mode {mode}: out rel-L2 {((a - ref).norm() / ref.norm()).item():.3e}  "
                  f"lse max {(res[mode][1].double() - lse_ref).abs().max().item():.3e}", flush=True)
    setopt(1)

if os.environ.get("WIN_TIME_ONLY") != "1":
    run(2, 4, (16, 20), ref64=True)
    run(3, 2, (30, 40), seed=1, ref64=True)
    run(1, 16, (30, 40), seed=2)
    run(5, 3, (7, 40), seed=3, ref64=True)       # ragged last chunk (7 rows = 2 chunks of 3 + 1)
    run(4, 3, (13, 20), seed=4, ref64=True)      # ragged (13 = 2 x 5 + 3), 261 tokens
if do_time:
    run(64, 16, (30, 40), seed=5, time_it=True)
