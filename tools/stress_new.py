"""Race screen for the kernels added late in round 1: repeated launches must reproduce bit-identical results where the
arithmetic is order-independent (binned rasterizer, streaming attention forward and dK/dV/dQ) and tolerance-identical
where fp32 atomics reorder sums (table gradient)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import datasets as D, ops
from oracle.vit_ref import rel_pos_index
g = torch.Generator(device="cuda").manual_seed(0)
# --- rasterizer
B, n, H, W = 24, 300_000, 480, 640
ev = torch.stack([torch.randint(0, W, (B * n,), generator=g, device="cuda").double(), torch.randint(0, H, (B * n,), generator=g, device="cuda").double(),
                  torch.rand((B * n,), generator=g, device="cuda", dtype=torch.float64), (torch.randint(0, 2, (B * n,), generator=g, device="cuda") * 2 - 1).double()], 1).contiguous()
off = torch.arange(0, B + 1, device="cuda", dtype=torch.int64) * n
ref = D.rasterize(ev, off, H, W, False, binned=True, strict=False)
for i in range(60):
    assert torch.equal(D.rasterize(ev, off, H, W, False, binned=True, strict=False), ref), i
print("rasterizer x60 ok")
# --- streaming attention
Bq, T, Hh, win = 3, 1201, 4, (30, 40)
Dm = 64 * Hh; TP = ops.attn_tokens_padded(T)
qkv = (torch.randn(Bq * T, 3 * Dm, generator=g, device="cuda") * 0.5).bfloat16()
idx, nrd = rel_pos_index(win)
table = torch.randn(nrd, Hh, generator=g, device="cuda") * 0.3
dout = torch.randn(Bq * T, Dm, generator=g, device="cuda").bfloat16()
def run():
    out = torch.zeros(Bq * T, Dm, dtype=torch.bfloat16, device="cuda"); lse = torch.zeros(Bq, Hh, TP, device="cuda")
    ops.attn_fwd(qkv, Bq, T, Dm, Hh, table, win, out, lse)
    delta = torch.zeros(2 * Bq * T + 4, Hh, device="cuda")
    ops.attn_delta(dout, out, Bq * T, Hh, delta)
    dqkv = torch.zeros(Bq * T, 3 * Dm, dtype=torch.bfloat16, device="cuda"); dt = torch.zeros(nrd, Hh, device="cuda"); dqb = torch.zeros(Dm, device="cuda")
    ops.attn_bwd(qkv, dout, lse, delta, table, win, Bq, T, Dm, Hh, 0.125, dqkv, dt, dq_bias=dqb)
    return out, lse, dqkv, dt
r0 = run()
for i in range(40):
    r = run()
    assert torch.equal(r[0], r0[0]) and torch.equal(r[1], r0[1]) and torch.equal(r[2], r0[2]), i
    assert (r[3] - r0[3]).abs().max() <= 1e-3 * r0[3].abs().max(), i
print("streaming attention x40 ok")
print("STRESS PASS")
