"""A/B of the 14x14 attention kernels (attn16.hip) against the general kernels (attn.hip) on the same inputs, plus timing."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops, _lib
from oracle.vit_ref import rel_pos_index

def setopt(v):
    assert _lib.lib.memhip_set_option(b"attn16", v) == 0

def run(B, H, seed=0, time_it=False):
    T, D = 197, 64 * H
    TP = ops.attn_tokens_padded(T)
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.7)
    qkv[:, :D] *= 0.5
    qkv = qkv.bfloat16()
    idx, nrd = rel_pos_index((14, 14))
    table = torch.randn(nrd, H, generator=g, device="cuda") * 0.5
    dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
    res = {}
    for mode in (0, 1):
        setopt(mode)
        out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
        dqkv = torch.full((B * T, 3 * D), 3.0, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
        delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
        ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)
        ops.attn_delta(dout, out, B * T, H, delta)
        ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, None)
        torch.cuda.synchronize()
        res[mode] = (out.float(), lse[:, :, :T].clone(), dqkv.float(), dtable.clone(), dqb.clone())
        if time_it:
            def t(f, n=10):
                f(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n): f()
                torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
            print(f"mode {mode}: fwd {t(lambda: ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)):.1f} us  "
                  f"bwd {t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, None)):.1f} us")
    names = ("out", "lse", "dqkv", "dtable", "dq_bias")
    for n, a, b in zip(names, res[0], res[1]):
        d = (a - b).abs().max().item(); rel = ((a - b).norm() / (a.norm() + 1e-30)).item()
        print(f"B={B} H={H} {n}: max|old-new| {d:.3e}  rel-L2 {rel:.3e}  max|old| {a.abs().max().item():.3e}  finite {bool(torch.isfinite(b).all())}")
    setopt(1)

def sweep():
    """timing of the 14x14 kernels over the stagger option (cycles)"""
    B, H, T = 256, 12, 197
    D = 64 * H; TP = ops.attn_tokens_padded(T)
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = (torch.randn(B * T, 3 * D, generator=g, device="cuda") * 0.5).bfloat16()
    idx, nrd = rel_pos_index((14, 14))
    table = torch.randn(nrd, H, generator=g, device="cuda") * 0.3
    dout = torch.randn(B * T, D, generator=g, device="cuda").bfloat16()
    out = torch.zeros(B * T, D, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(B, H, TP, device="cuda")
    dqkv = torch.zeros(B * T, 3 * D, dtype=torch.bfloat16, device="cuda"); dtable = torch.zeros(nrd, H, device="cuda")
    delta = torch.zeros(2 * B * T + 4, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
    ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse); ops.attn_delta(dout, out, B * T, H, delta)
    def t(f, n=20):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    for st in (40000, 0, 20000, 0, 30000, 40000, 50000):
        assert _lib.lib.memhip_set_option(b"attn16_stagger", st) == 0
        assert _lib.lib.memhip_set_option(b"attn16_stagger_fwd", st) == 0
        print(f"stagger {st:6d}: fwd {t(lambda: ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)):.1f} us  "
              f"bwd {t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dtable, dqb, None)):.1f} us  "
              f"bwd(no dtable) {t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, None, dqb, None)):.1f} us")
    _lib.lib.memhip_set_option(b"attn16_stagger", 40000); _lib.lib.memhip_set_option(b"attn16_stagger_fwd", 0)

run(3, 12)
run(45, 4, seed=1)
run(256, 12, seed=2, time_it=True)
sweep()
