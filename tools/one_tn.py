"""One weight-gradient GEMM (and one NT GEMM) for counter runs: python tools/one_tn.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mem_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
R, N, K = 50432, 3072, 768
A = torch.randn(R, N, device="cuda").bfloat16(); B = torch.randn(R, K, device="cuda").bfloat16()
o = torch.zeros(N, K, device="cuda")
X = torch.randn(R, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
y = torch.empty(R, N, dtype=torch.bfloat16, device="cuda")
for _ in range(reps):
    ops.gemm_tn(A, B, R, N, K, o, accumulate=True)
    ops.gemm_nt(X, W, R, N, K, ops.EPI_BIAS_BF16, out0=y)
torch.cuda.synchronize()
