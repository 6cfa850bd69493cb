"""Host-side product functions against the reference's outputs (no GPU): the functions a training run actually
calls -- mem_amd.utils.cosine_scheduler (utils.py:395-412), mem_amd.optim_factory.get_parameter_groups
(optim_factory.py:56-100) on the PRODUCT model -- bound to the fixtures oracle/gen_golden.py wrote from the imported
reference (schedules.npz, vit_meta.json)."""
import contextlib
import io
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TINY = dict(img_size=(64, 64), patch_size=(16, 16), in_chans=3, vocab_size=512, embed_dim=128, depth=2,
            num_heads=2, mlp_ratio=4, drop_path_rate=0.0, use_shared_rel_pos_bias=True,
            use_abs_pos_emb=False, init_values=0.1)


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def test_product_cosine_scheduler_equals_reference_arrays():
    from mem_amd.utils import cosine_scheduler
    g = np.load(os.path.join(GOLDEN, "schedules.npz"))
    s1 = _quiet(cosine_scheduler, 5e-4, 1e-5, 3000, 8, warmup_epochs=5, warmup_steps=1000)      # ncaltech.conf lr
    assert s1.dtype == np.float64 and len(s1) == 3000 * 8
    assert np.array_equal(s1[:1200], g["lr_ncaltech_head"]) and np.array_equal(s1[-200:], g["lr_ncaltech_tail"])
    s2 = _quiet(cosine_scheduler, 0.05, 0.05, 3000, 8)                                           # weight-decay schedule
    assert np.array_equal(s2[:16], g["wd"])
    s3 = _quiet(cosine_scheduler, 5e-4, 1e-5, 2, 10, warmup_epochs=5, warmup_steps=4)
    assert np.array_equal(s3, g["lr_small"])
    # warmup_epochs without warmup_steps: warmup_iters = warmup_epochs * niter_per_ep (utils.py:398-400)
    s4 = _quiet(cosine_scheduler, 1.0, 0.0, 4, 5, warmup_epochs=1)
    assert np.array_equal(s4[:5], np.linspace(0, 1.0, 5)) and s4[5] == 1.0 and len(s4) == 20


def test_product_parameter_groups_equal_reference_membership():
    """The product's pt_vit (constructed on CPU: the fused engine is created lazily) through the product's
    get_parameter_groups / create_optimizer argument path: same state-dict keys and shapes, same decay / no_decay
    membership IN THE SAME ORDER as the reference's create_optimizer produced (tests/golden/vit_meta.json)."""
    from mem_amd import optim_factory as OF
    from mem_amd.modeling_pretrain import create_model
    meta = json.load(open(os.path.join(GOLDEN, "vit_meta.json")))
    torch.manual_seed(0)
    m = create_model("pt_vit", pretrained=False, drop_block_rate=None, **TINY)
    sd = m.state_dict()
    assert list(sd.keys()) == meta["tiny_state_keys"]
    assert {k: list(v.shape) for k, v in sd.items()} == meta["tiny_state_shapes"]
    groups = _quiet(OF.get_parameter_groups, m, 0.05, m.no_weight_decay())
    name_of = {id(p): n for n, p in m.named_parameters()}
    got = {("no_decay" if g["weight_decay"] == 0 else "decay"): [name_of[id(p)] for p in g["params"]] for g in groups}
    assert got == meta["tiny_groups"]
    assert [g["weight_decay"] for g in groups] == [0.0, 0.05] and all(g["lr_scale"] == 1.0 for g in groups)
    assert "mask_token" in got["decay"] and "cls_token" in got["no_decay"]       # the reference's skip list quirk


def test_product_vit_base_parameter_inventory():
    """ViT-B (BASELINE configs[1], C=2) and C=3: parameter names, count and group sizes as the reference's."""
    from mem_amd import optim_factory as OF
    from mem_amd.modeling_pretrain import pt_vit
    meta = json.load(open(os.path.join(GOLDEN, "vit_meta.json")))
    for c in (2, 3):
        m = pt_vit(img_size=(224, 224), patch_size=(16, 16), in_chans=c, vocab_size=8192, embed_dim=768, depth=12,
                   num_heads=12, mlp_ratio=4, drop_path_rate=0.1, use_shared_rel_pos_bias=True, use_abs_pos_emb=False,
                   init_values=0.1)
        assert [n for n, _ in m.named_parameters()] == meta[f"base_c{c}_param_names"]
        assert sum(p.numel() for p in m.parameters()) == meta[f"base_c{c}_nparams"]
        groups = _quiet(OF.get_parameter_groups, m, 0.05, m.no_weight_decay())
        counts = {("no_decay" if g["weight_decay"] == 0 else "decay"): len(g["params"]) for g in groups}
        # per block 4 matrices / 11 vectors and biases; + mask_token, patch-embed kernel, rel-pos table, lm_head.weight
        assert counts == {"no_decay": 12 * 11 + 5, "decay": 12 * 4 + 4}


def test_cli_defaults_match_reference():
    """run_mem_pretraining.py:152-157 defaults the driver relies on when a config omits them."""
    from mem_amd.run_mem_pretraining import get_args
    a = _quiet(get_args, ["--expweek", "t"])
    assert a.num_workers == 10 and a.pin_mem is True and a.synthetic_if_missing == 0
    assert a.model == "pt_vit" and a.opt == "adamw" and a.clip_grad is None


def test_missing_data_path_fails_like_the_reference():
    """datasets.py:149-154 asserts the dataset root exists.  A typo must not silently train on synthetic streams."""
    import pytest
    from mem_amd.datasets import build_pretraining_dataset
    from mem_amd.run_mem_pretraining import get_args
    def get(*extra):
        a = _quiet(get_args, ["--expweek", "t", *extra])
        a.window_size = (a.input_H // 16, a.input_W // 16)          # main() sets it from the model (run_mem_pretraining.py:251)
        return a
    for path in ("/data/mydset", "/nonexistent/ncaltech101/"):
        a = get("--data_path", path)
        with pytest.raises(AssertionError, match="not found"):
            _quiet(build_pretraining_dataset, a)
    buf = io.StringIO()
    a = get("--data_path", "/data/mydset", "--synthetic_if_missing", "1")
    with contextlib.redirect_stdout(buf):
        ds = build_pretraining_dataset(a)
    assert "WARNING" in buf.getvalue() and "SYNTHETIC" in buf.getvalue() and len(ds) == 64      # warns also without a sensor keyword
    a = get("--data_path", "synthetic")
    assert len(_quiet(build_pretraining_dataset, a)) == 64


def test_host_thread_cap_respects_the_container_quota(monkeypatch):
    """cap_host_threads: never more threads than the CFS quota divided by the rank processes on the node, never raises
    the pool (the launch thread was throttled for 80 ms at a time when torch's 128-thread pool burnt the 16-CPU quota)."""
    import os
    import torch
    from mem_amd.utils import cap_host_threads, host_cpu_budget
    before = torch.get_num_threads()
    try:
        budget = host_cpu_budget()
        assert 1 <= budget <= (os.cpu_count() or 1)
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
        n = cap_host_threads(4)
        assert 1 <= n <= max(1, min(4, budget // 8, before)) and torch.get_num_threads() <= before
        monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
        assert cap_host_threads(64) <= torch.get_num_threads()          # a larger limit does not raise the pool again
    finally:
        torch.set_num_threads(before)
