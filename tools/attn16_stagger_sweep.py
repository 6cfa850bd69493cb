import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mem_amd import ops, _lib
B, T, H, D = 256, 197, 12, 768
M = B * T
torch.manual_seed(0)
qkv = (torch.randn(M, 3 * D, device="cuda") * 0.5).bfloat16()
table = torch.randn(732, H, device="cuda") * 0.1
out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, ops.attn_tokens_padded(T), device="cuda")
dout = (torch.randn(M, D, device="cuda") * 0.1).bfloat16()
delta = torch.zeros((2 * M + 4) * H, device="cuda")
dqkv = torch.zeros(M, 3 * D, device="cuda", dtype=torch.bfloat16); dt = torch.zeros(732, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
def t(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)
for st in (0, 10000, 20000, 40000, 60000, 90000):
    _lib.set_option("attn16_stagger", st)
    b = t(lambda: ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dt, dq_bias=dqb, out=out))
    print(f"attn16_stagger {st:6d}: bwd (fused delta) {b:6.1f} us", flush=True)
_lib.set_option("attn16_stagger", 40000)
for st in (0, 10000, 20000, 40000):
    _lib.set_option("attn16_stagger_fwd", st)
    f = t(lambda: ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse))
    print(f"attn16_stagger_fwd {st:6d}: fwd {f:6.1f} us", flush=True)
