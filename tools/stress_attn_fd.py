import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mem_amd import ops
T, H, D = 197, 12, 768
for B in (256, 40, 7):
    M = B * T
    torch.manual_seed(B)
    qkv = (torch.randn(M, 3 * D, device="cuda") * 0.5).bfloat16()
    table = torch.randn(732, H, device="cuda") * 0.1
    out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, ops.attn_tokens_padded(T), device="cuda")
    ops.attn_fwd(qkv, B, T, D, H, table, (14, 14), out, lse)
    dout = (torch.randn(M, D, device="cuda") * 0.1).bfloat16()
    delta = torch.zeros((2 * M + 4) * H, device="cuda")
    ref = None; bad = 0
    n = 300 if B < 256 else 120
    for it in range(n):
        dqkv = torch.full((M, 3 * D), 5.0, device="cuda", dtype=torch.bfloat16); dt = torch.zeros(732, H, device="cuda"); dqb = torch.zeros(D, device="cuda")
        ops.attn_bwd(qkv, dout, lse, delta, table, (14, 14), B, T, D, H, 0.125, dqkv, dt, dq_bias=dqb, out=out)
        if ref is None: ref = dqkv.clone()
        elif not torch.equal(ref, dqkv): bad += 1
    torch.cuda.synchronize()
    print(f"B={B}: {n} launches, {bad} differ from the first", flush=True)
