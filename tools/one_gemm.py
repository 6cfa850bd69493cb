import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
m, n, k = [int(v) for v in sys.argv[1:4]]
A = torch.randn(m, k, device="cuda").bfloat16(); B = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
o = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
for _ in range(5):
    ops.gemm_nt(A, B, m, n, k, ops.EPI_BIAS_BF16, out0=o)
torch.cuda.synchronize()
