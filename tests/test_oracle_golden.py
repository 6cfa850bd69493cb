"""-m "not gpu": the oracle restatement reproduces the committed reference goldens
(tests/golden/*, generated from the imported reference by oracle/gen_golden.py)."""
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import events_np as E
from oracle import masking_py as MP
from oracle import transforms_t as T
from oracle import vit_ref as V


def test_rasterizer_goldens():
    g = np.load(os.path.join(GOLDEN, "events_raster.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "events_raster.json")))
    assert len(meta) >= 12
    for m in meta:
        img = E.event_arr_to_img(g[m["name"] + "__ev"], m["H"], m["W"], m["timesurface"])
        assert img.dtype == np.uint8 and np.array_equal(img, g[m["name"] + "__img"]), m["name"]
    ev = g["wrap256__ev"]                                      # uint8 wrap-around is exercised
    hits = int(((ev[:, 0] == 7) & (ev[:, 1] == 5) & (ev[:, 3] == 1)).sum())
    assert hits > 255 and g["wrap256__img"][5, 7, 0] == hits % 256
    with pytest.raises(IndexError):
        E.event_arr_to_img(g["oob__ev"], 100, 120, False)


def test_event_aug_goldens():
    g = np.load(os.path.join(GOLDEN, "events_augs.npz"))
    small = g["small__in"]
    assert np.array_equal(E.slice_random_max_evs(g["slice__in"], 30000, int(g["slice__start"])), g["slice__out"])
    for tag in ("flip", "noflip"):
        u = float(g[f"timeflip_{tag}__u"])
        assert np.array_equal(E.random_time_flip(small.copy(), u), g[f"timeflip_{tag}__out"])
        assert np.array_equal(E.flip_along_x(small, u), g[f"flipx_{tag}__out"])
        assert np.array_equal(E.flip_along_x(small, u, W=300), g[f"flipx300_{tag}__out"])
    for tag, HW in (("a", (None, None)), ("b", (180, 240)), ("c", (None, None))):
        xs, ys = g[f"shift_{tag}__xy"]
        assert np.array_equal(E.random_shift(small, xs, ys, HW[0], HW[1]), g[f"shift_{tag}__out"])
    for tr in (0, 1):
        assert np.array_equal(E.reshape_scale_xy(g[f"rescale_{tr}__in"], 224, 224, 480, 640, bool(tr)),
                              g[f"rescale_{tr}__out"])


def test_ncaltech_decode_golden():
    g = np.load(os.path.join(GOLDEN, "ncaltech_records.npz"))
    assert np.array_equal(E.decode_ncaltech101(g["raw"].tobytes()), g["events"])


def test_record_goldens_from_reference():
    """oracle/gen_golden_records.py: outputs of the reference's own ncaltech101() / imgnet_npy_loader / dsec_npy_loader."""
    g = np.load(os.path.join(GOLDEN, "records.npz"))
    names = sorted({k.split("__")[1] for k in g.files if k.startswith("ncaltech__")})
    assert len(names) == 4
    for n in names:
        assert np.array_equal(E.decode_ncaltech101(g[f"ncaltech__{n}__raw"].tobytes()), g[f"ncaltech__{n}__events"])
    for tag in ("u16_i64_bool", "i32_f64_u8", "i16_u32_i16wrap"):
        got = E.imgnet_struct_to_events(*(g[f"imgnet__{tag}__{k}"] for k in "xytp"))
        assert np.array_equal(got, g[f"imgnet__{tag}__events"])
    rec = g["imgnet__struct__rec"]
    assert np.array_equal(E.imgnet_struct_to_events(rec["x"], rec["y"], rec["t"], rec["p"]), g["imgnet__struct__events"])
    for tag in ("f64", "i64", "u16"):
        assert np.array_equal(E.dsec_to_events(g[f"dsec__{tag}__in"]), g[f"dsec__{tag}__events"])


def test_transform_goldens():
    g = np.load(os.path.join(GOLDEN, "transforms.npz"))
    for name in ("s32", "s224", "zeros"):
        x = torch.from_numpy(g[name + "__in"])
        r = T.remove_timesurface(x)
        assert np.array_equal(r.numpy(), g[name + "__rm_ts"])
        assert np.array_equal(T.remove_hot_pixels(r, 10.0).numpy(), g[name + "__hot10"])
        assert np.array_equal(T.remove_hot_pixels(r, 3.0).numpy(), g[name + "__hot3"])
        n = T.normalize_event(T.remove_hot_pixels(r, 10.0))
        assert np.array_equal(n.numpy(), g[name + "__norm"])
        assert np.array_equal(T.event_chain(x).numpy(), g[name + "__norm"])
        assert np.array_equal(T.log_transform(r).numpy(), g[name + "__log"])
        assert np.array_equal(T.gamma_transform(r, 0.5).numpy(), g[name + "__gamma"])
        assert np.array_equal(T.to_uint8(n).numpy(), g[name + "__u8"])
        assert np.array_equal(T.to_float32(T.to_uint8(n)).numpy(), g[name + "__f32"])
    assert (g["s224__hot3"] != g["s224__rm_ts"]).any()           # the hot-pixel branch fires


def test_mask_goldens():
    g = np.load(os.path.join(GOLDEN, "masks.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "masks.json")))
    for c in meta["cfgs"]:
        for s in meta["seeds"][:3]:
            random.seed(s)
            o = MP.BlockMaskOracle(tuple(c["size"]), c["n"], min_num_patches=c["lo"], max_num_patches=c["hi"])
            got = np.stack([o() for _ in range(meta["per_seed"])])
            want = np.unpackbits(g[f"{c['tag']}__s{s}"], axis=1)[:, :got[0].size].reshape(got.shape)
            assert np.array_equal(got, want), (c, s)
    random.seed(12345)
    assert np.array_equal(np.array([random.random() for _ in range(16)]), g["mt__random_s12345"])


def test_hot_pixel_topk_oracle_vs_reference_golden():
    """RemoveHotPixels(num_hot_pixels=k): the oracle restatement against the reference's outputs (tie-free boundaries)."""
    from oracle import transforms_t as OT
    g = np.load(os.path.join(GOLDEN, "transforms_topk.npz"))
    for name in ("t32", "t224", "t40x56"):
        x = torch.from_numpy(g[name + "__in"])
        for k in g[name + "__ks"].tolist():
            assert OT.topk_is_tie_free(x, k)
            assert np.array_equal(OT.remove_hot_pixels_topk(x, k).numpy(), g[name + f"__top{k}"]), (name, k)
    x = torch.from_numpy(g["t32__in"])
    assert OT.topk_clamped(x, 10 ** 6) == int(x[0::2].sum() / 4) < 100         # the sum / 4 clamp was exercised


def test_schedule_goldens():
    g = np.load(os.path.join(GOLDEN, "schedules.npz"))
    s = V.cosine_scheduler(5e-4, 1e-5, 3000, 8, warmup_epochs=5, warmup_steps=1000)
    assert np.array_equal(s[:1200], g["lr_ncaltech_head"]) and np.array_equal(s[-200:], g["lr_ncaltech_tail"])
    assert np.array_equal(V.cosine_scheduler(5e-4, 1e-5, 2, 10, warmup_epochs=5, warmup_steps=4), g["lr_small"])


TINY = dict(img_size=(64, 64), patch_size=(16, 16), in_chans=3, vocab_size=512, embed_dim=128, depth=2,
            num_heads=2, mlp_ratio=4, drop_path_rate=0.0, use_shared_rel_pos_bias=True,
            use_abs_pos_emb=False, init_values=0.1)


def test_vit_tiny_fwdbwd_golden():
    torch.set_num_threads(1)
    g = np.load(os.path.join(GOLDEN, "vit_tiny_fwdbwd.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "vit_meta.json")))
    m = V.RefViT(**TINY)
    assert list(m.state_dict().keys()) == meta["tiny_state_keys"]
    m.load_state_dict(V.fill_by_name(m.state_dict(), seed=0))
    x, mask, labels = torch.from_numpy(g["x"]), torch.from_numpy(g["mask"]), torch.from_numpy(g["labels"])
    lo = m(x, mask)
    loss = torch.nn.CrossEntropyLoss()(lo, labels)
    loss.backward()
    assert np.array_equal(lo.detach().numpy(), g["fp32__logits"])
    assert np.array_equal(loss.detach().numpy(), g["fp32__loss"])
    for k, p in m.named_parameters():
        assert np.array_equal(p.grad.numpy(), g[f"fp32__grad__{k}"]), k
    _, names = V.param_groups(m)
    assert names["no_decay"] == meta["tiny_groups"]["no_decay"] and names["decay"] == meta["tiny_groups"]["decay"]
    assert "mask_token" in names["decay"] and "cls_token" in names["no_decay"]


def test_vit_long_nonsquare_golden():
    """256 x 320 canvas (16 x 20 + 1 = 321 tokens, non-square relative-position window): the oracle against
    the reference outputs of oracle/gen_golden_long.py, bit for bit (bf16 autocast incl. every gradient)."""
    from oracle.gen_golden import vit_inputs
    from oracle.gen_golden_long import LONG, LONG_INPUTS
    torch.set_num_threads(1)
    g = np.load(os.path.join(GOLDEN, "vit_long.npz"))
    m = V.RefViT(**LONG)
    m.load_state_dict(V.fill_by_name(m.state_dict(), seed=3))
    x, mask, labels = vit_inputs(LONG, *LONG_INPUTS)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lo = m(x, mask)
        loss = torch.nn.CrossEntropyLoss()(lo, labels)
    loss.backward()
    assert np.array_equal(lo.detach().float().numpy(), g["bf16__logits"])
    assert np.array_equal(loss.detach().numpy(), g["bf16__loss"])
    for k, p in m.named_parameters():
        assert np.array_equal(p.grad.numpy(), g[f"bf16__grad__{k}"]), k
    with torch.no_grad():
        assert np.array_equal(m(x, mask).numpy(), g["fp32__logits"])


def test_config5_geometry_golden_oracle_fp32():
    """tests/golden/vit_c5.npz (reference, BASELINE configs[4] geometry at depth 2): the oracle's fp32 forward reproduces the
    reference's loss and logits (to 1e-5: the fixture was written with 8 CPU threads; the generator asserts bit-equality
    in-process, gradients included, in fp32 and under bf16 autocast)."""
    from oracle.gen_golden import vit_inputs
    from oracle.gen_golden_c5 import C5, C5_INPUTS
    g = np.load(os.path.join(GOLDEN, "vit_c5.npz"))
    m = V.RefViT(**C5)
    m.load_state_dict(V.fill_by_name(m.state_dict(), seed=9))
    assert tuple(m.state_dict()["rel_pos_bias.relative_position_bias_table"].shape) == (4664, 16)
    x, mask, labels = vit_inputs(C5, *C5_INPUTS)
    with torch.no_grad():
        lo = m(x, mask)
        loss = torch.nn.CrossEntropyLoss()(lo, labels)
    assert abs(float(loss) - float(g["fp32__loss"])) <= 1e-5
    assert np.abs(lo[:96].numpy() - g["fp32__logits_head"]).max() <= 1e-4


def test_finetune_model_golden_and_layer_decay_groups():
    """f3: the oracle's finetuning model (RefFtViT) against the reference outputs of oracle/gen_golden_ft.py, bit for
    bit under bf16 autocast, in both head configurations; the layer-decay parameter groups of the oracle AND of the
    product's optim_factory (pure host logic) against the reference's assignment."""
    from oracle.gen_golden_ft import FT_A, FT_B, ft_inputs
    torch.set_num_threads(1)
    g = np.load(os.path.join(GOLDEN, "vit_ft.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "vit_ft_meta.json")))
    for tag, cfg in (("a", FT_A), ("b", FT_B)):
        m = V.RefFtViT(**cfg)
        assert list(m.state_dict().keys()) == meta[f"{tag}_state_keys"]
        m.load_state_dict(V.fill_by_name(m.state_dict(), seed=5))
        x, y = ft_inputs(cfg, 5, 31)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lo = m(x)
            loss = torch.nn.CrossEntropyLoss()(lo, y)
        loss.backward()
        assert np.array_equal(lo.detach().float().numpy(), g[f"{tag}__logits"])
        assert np.array_equal(loss.detach().numpy(), g[f"{tag}__loss"])
        for k, p in m.named_parameters():
            assert np.array_equal(p.grad.numpy(), g[f"{tag}__grad__{k}"]), (tag, k)
        want = meta[f"{tag}_layer_decay_groups"]
        assert V.layer_decay_groups(m, 0.05, 0.75) == want
        # the product's host-side group builder (no GPU needed: it only looks at names and shapes)
        import contextlib
        import io
        from mem_amd import optim_factory as OF
        depth = cfg["depth"]
        assigner = OF.LayerDecayValueAssigner(list(0.75 ** (depth + 1 - i) for i in range(depth + 2)))
        name_of = {id(p): n for n, p in m.named_parameters()}
        with contextlib.redirect_stdout(io.StringIO()):
            groups = OF.get_parameter_groups(m, 0.05, m.no_weight_decay(), assigner.get_layer_id, assigner.get_scale)
        got = {json.dumps([name_of[id(p)] for p in gr["params"]]): (gr["weight_decay"], gr["lr_scale"]) for gr in groups}
        assert got == {json.dumps(v["params"]): (v["weight_decay"], v["lr_scale"]) for v in want.values()}


def test_vit_tiny_train_golden():
    """First 20 of the 100 golden steps (the full 100 are checked against the HIP path on the GPU)."""
    torch.set_num_threads(1)
    from oracle.gen_golden import vit_inputs
    g = np.load(os.path.join(GOLDEN, "vit_tiny_train100.npz"))
    m = V.RefViT(**TINY)
    m.load_state_dict(V.fill_by_name(m.state_dict(), seed=0))
    opt = V.make_optimizer(m)
    for it in range(20):
        xb, mb, lb = vit_inputs(TINY, 4, 1000 + it % 8, 6)
        loss, gn, acc = V.train_step(m, opt, xb, mb, lb, it, g["lr"], g["wd"], clip_grad=30.0)
        assert loss == g["fp32__loss"][it] and gn == g["fp32__gnorm"][it], it


def test_vae_tokenizer_oracle_and_module_vs_reference_golden():
    """tests/golden/vae_tiny.npz was written by the REFERENCE DiscreteVAE (oracle/gen_golden_vae.py):
    the oracle restatement and the product's torch module must reproduce its logits and ids exactly."""
    import os
    import numpy as np
    import torch
    from oracle.vae_ref import TINY_VAE, encoder_logits, fill_vae_by_name, get_codebook_indices, vae_inputs
    from mem_amd.vae_model import DiscreteVAE
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vae_tiny.npz"))
    torch.set_num_threads(1)
    m = DiscreteVAE(**TINY_VAE).eval()
    sd = fill_vae_by_name(m.state_dict(), seed=0)
    assert sorted(sd.keys()) == list(g["keys"])                   # same module tree / checkpoint keys
    m.load_state_dict(sd)
    img = vae_inputs(TINY_VAE, 6, 11)
    with torch.no_grad():
        lo = encoder_logits(sd, img, TINY_VAE["num_layers"], TINY_VAE["num_resnet_blocks"])
        ids = get_codebook_indices(sd, img, TINY_VAE["num_layers"], TINY_VAE["num_resnet_blocks"])
        assert np.array_equal(lo[0].numpy(), g["logits_b0"])
        assert np.array_equal(ids.numpy().astype(np.int32), g["ids"])
        assert np.array_equal(m.get_codebook_indices(img).numpy().astype(np.int32), g["ids"])
        assert torch.equal(m(img, return_logits=True), lo)
