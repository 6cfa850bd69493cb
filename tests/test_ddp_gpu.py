"""-m gpu: data-parallel equivalence ("1 vs N ranks, same global batch => same loss").
Reference behaviour: DistributedSampler split + DDP gradient mean (mem/run_mem_pretraining.py:302-309,365-367),
per-rank token-mean loss (mem/engine_for_pretraining.py:152).  With equal masked-token counts on every rank the
mean of the per-rank gradients IS the gradient of the global-batch loss, so a 2-rank job must reproduce the
single-rank job on the concatenated batch up to bf16 accumulation-order noise.

Two forms, both on the ONE GPU of the test box:
  * two real rank processes (tests/ddp_worker.py) running the product's N>1 path end to end (process group,
    parameter broadcast, per-bucket async all-reduce from the backward hook, join, clip, AdamW) -- gloo on CUDA
    tensors, because RCCL refuses two ranks on one device;
  * the same split run sequentially in one process with the reducer's arithmetic applied by hand."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

STEPS, PER_RANK = 3, 3
# bf16 GEMM operands: a split batch changes which rows share a tile / the order of fp32 atomics, nothing else
LOSS_TOL, PARAM_REL_TOL = 2e-3, 2e-2


def _single():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_worker as W
    m, opt, batches = W.make_job(2, None, STEPS, PER_RANK, seed_w=3)
    rec = W.run_steps(m, opt, batches)
    return rec, m.engine.flat_p.clone(), W


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_processes_match_single_rank_global_batch(tmp_path):
    rec1, p1, _ = _single()
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    outs = [str(tmp_path / f"rank{r}.pt") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), "--rank", str(r), "--world", "2",
                               "--port", str(port), "--steps", str(STEPS), "--per-rank", str(PER_RANK), "--out", outs[r]],
                              env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    r0, r1 = torch.load(outs[0]), torch.load(outs[1])
    assert torch.equal(r0["flat_p"], r1["flat_p"]), "ranks diverged: broadcast / all-reduce did not keep replicas equal"
    loss2 = (np.array(r0["rec"])[:, 0] + np.array(r1["rec"])[:, 0]) / 2
    loss1 = np.array(rec1)[:, 0]
    print("loss 1 rank", loss1, " mean of 2 ranks", loss2)
    assert np.abs(loss1 - loss2).max() <= LOSS_TOL
    # identical post-reduce gradient on both ranks => identical reported norm, equal to the single-rank norm
    assert r0["rec"][-1][1] == r1["rec"][-1][1]
    assert abs(r0["rec"][0][1] / rec1[0][1] - 1) <= 2e-2
    d = (r0["flat_p"].cuda() - p1).norm() / (p1 - _initial_params()).norm()
    print("relative difference of the 3-step parameter update: %.3e" % float(d))
    assert float(d) <= PARAM_REL_TOL


def _initial_params():
    import ddp_worker as W
    m, _, _ = W.make_job(2, None, 0, PER_RANK, seed_w=3)
    return m.engine.flat_p.clone()


def test_sequential_split_with_manual_gradient_mean():
    """One step: g(rank 0 half), g(rank 1 half) from two engines, averaged by hand == the B = 2b gradient."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_worker as W
    m, _, b = W.make_job(2, None, 1, PER_RANK, seed_w=3)
    x, mask, lab, dp = b[0]
    la = m.forward_loss(x.cuda(), mask.cuda(), lab.cuda(), drop_path_masks=dp.cuda()); m.backward()
    g_full, l_full = m.engine.flat_g.clone(), float(la[0])
    gs, ls = [], []
    for r in range(2):
        mr, _, br = W.make_job(2, r, 1, PER_RANK, seed_w=3 - r)      # seed_w + rank == 3: the broadcast weights
        xr, maskr, labr, dpr = br[0]
        lr_ = mr.forward_loss(xr.cuda(), maskr.cuda(), labr.cuda(), drop_path_masks=dpr.cuda()); mr.backward()
        gs.append(mr.engine.flat_g.clone()); ls.append(float(lr_[0]))
    g_mean = (gs[0] + gs[1]) / 2
    rel = float((g_mean - g_full).norm() / g_full.norm())
    cos = float(torch.dot(g_mean, g_full) / (g_mean.norm() * g_full.norm()))
    print("loss %.6f vs %.6f; grad rel-L2 %.3e cos %.6f" % (l_full, sum(ls) / 2, rel, cos))
    assert abs(l_full - sum(ls) / 2) <= 1e-3 and rel <= 2e-2 and cos >= 0.9995
