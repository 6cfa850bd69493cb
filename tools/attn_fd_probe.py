"""Fused delta in the 14 x 14 attention backward: results and time against attn_delta + attn_bwd (B = 256, 12 heads)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
B, T, H, D = int(os.environ.get("B", 256)), 197, 12, 768
M = B * T
torch.manual_seed(0)
qkv = (torch.randn(M, 3 * D, device="cuda") * 0.5).bfloat16()
table = torch.randn(732, H, device="cuda") * 0.1
out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, ops.attn_tokens_padded(T), device="cuda")
ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)
dout = (torch.randn(M, D, device="cuda") * 0.1).bfloat16()
delta = torch.zeros((2 * M + 4) * H, device="cuda")
def old(dqkv, dt, dqb):
    ops.attn_delta(dout, out, M, H, delta)
    ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dt, dq_bias=dqb)
def new(dqkv, dt, dqb):
    ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dt, dq_bias=dqb, out=out)
res = []
for f in (old, new):
    dqkv = torch.zeros(M, 3 * D, device="cuda", dtype=torch.bfloat16); dt = torch.zeros(732, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
    f(dqkv, dt, dqb); torch.cuda.synchronize(); res.append((dqkv.float(), dt.clone(), dqb.clone()))
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-20))
print("dqkv rel-L2 %.2e  max|d| %.3e  equal %.4f | dtable rel %.2e | dq_bias rel %.2e" % (
    rel(res[1][0], res[0][0]), float((res[1][0] - res[0][0]).abs().max()), float((res[1][0] == res[0][0]).float().mean()),
    rel(res[1][1], res[0][1]), rel(res[1][2], res[0][2])))
def t(f, n=20):
    dqkv = torch.zeros(M, 3 * D, device="cuda", dtype=torch.bfloat16); dt = torch.zeros(732, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
    for _ in range(3): f(dqkv, dt, dqb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f(dqkv, dt, dqb)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    print("attn_delta + attn_bwd %.1f us   fused %.1f us" % (t(old), t(new)), flush=True)
# without the table gradient (the MAE decoder: zero table, no dtable)
def t2(f, n=20):
    dqkv = torch.zeros(M, 3 * D, device="cuda", dtype=torch.bfloat16); dqb = torch.zeros(D, device="cuda")
    for _ in range(3): f(dqkv, None, dqb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f(dqkv, None, dqb)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print("no table gradient: attn_delta + attn_bwd %.1f us   fused entry point %.1f us" % (t2(old), t2(new)), flush=True)
