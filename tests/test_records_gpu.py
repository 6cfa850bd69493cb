"""-m gpu: the event record contract on the HIP path (mem_amd/process_data.py -> csrc/records.hip) against the outputs
of the REFERENCE's own decoder / loaders (tests/golden/records.npz, oracle/gen_golden_records.py) -- bit-exact -- and
against the oracle on larger seeded inputs (ragged sizes around the 256-record / 1024-row workgroup boundaries).
Reference: process_data/process_dataset.py:48-63, mem/dataset_folder.py:275-292."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import events_np as E

pytestmark = pytest.mark.gpu


def test_ncaltech_decode_vs_reference_golden(tmp_path):
    from mem_amd import process_data as PD
    g = np.load(os.path.join(GOLDEN, "records.npz"))
    for n in sorted({k.split("__")[1] for k in g.files if k.startswith("ncaltech__")}):
        raw, want = g[f"ncaltech__{n}__raw"], g[f"ncaltech__{n}__events"]
        got = PD.decode_ncaltech101(raw.tobytes())
        assert got.dtype == torch.float64 and tuple(got.shape) == want.shape
        assert np.array_equal(got.cpu().numpy(), want), n
        f = tmp_path / (n + ".bin")
        raw.tofile(f)
        assert np.array_equal(PD.ncaltech101_file(str(f)).cpu().numpy(), want)
    hand = np.load(os.path.join(GOLDEN, "ncaltech_records.npz"))
    assert np.array_equal(PD.decode_ncaltech101(hand["raw"]).cpu().numpy(), hand["events"])
    with pytest.raises(IndexError):
        PD.decode_ncaltech101(bytes(7))


@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 1279, 100003])
def test_ncaltech_decode_vs_oracle_sizes(n):
    from mem_amd import process_data as PD
    raw = np.random.default_rng(n).integers(0, 256, n * 5, dtype=np.uint8)
    assert np.array_equal(PD.decode_ncaltech101(raw).cpu().numpy(), E.decode_ncaltech101(raw.tobytes()))


def test_imgnet_and_dsec_loaders_vs_reference_golden(tmp_path):
    from mem_amd import process_data as PD
    g = np.load(os.path.join(GOLDEN, "records.npz"))
    for tag in ("u16_i64_bool", "i32_f64_u8", "i16_u32_i16wrap"):
        cols = {k: g[f"imgnet__{tag}__{k}"] for k in "xytp"}
        path = str(tmp_path / (tag + ".npz"))
        np.savez(path, **cols)
        assert np.array_equal(PD.imgnet_npy_loader(path).cpu().numpy(), g[f"imgnet__{tag}__events"]), tag
    path = str(tmp_path / "struct.npy")
    np.save(path, g["imgnet__struct__rec"])
    assert np.array_equal(PD.imgnet_npy_loader(path).cpu().numpy(), g["imgnet__struct__events"])
    for tag in ("f64", "i64", "u16"):
        path = str(tmp_path / f"dsec_{tag}.npy")
        np.save(path, g[f"dsec__{tag}__in"])
        got = PD.dsec_npy_loader(path)
        assert np.array_equal(got.cpu().numpy(), g[f"dsec__{tag}__events"]), tag


@pytest.mark.parametrize("n", [0, 1, 1023, 1024, 1025, 300007])
def test_dsec_compaction_vs_oracle_sizes(n):
    from mem_amd import process_data as PD
    r = np.random.default_rng(n + 5)
    d = np.stack([r.integers(0, 640, n), r.integers(380, 480, n), np.sort(r.integers(0, 10**6, n)), r.integers(0, 2, n)], 1)
    d = d.astype(np.float64).reshape(n, 4)
    want = E.dsec_to_events(d.copy())
    got = PD.dsec_rows_to_events(d).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    # all rows kept / all rows dropped
    assert PD.dsec_rows_to_events(d, y_limit=1e9).shape[0] == n and PD.dsec_rows_to_events(d, y_limit=-1.0).shape[0] == 0


def test_decoded_events_feed_the_rasterizer():
    """The decoded rows are the rasterizer's input layout: N-Caltech-like records -> histogram == oracle chain."""
    from mem_amd import datasets as D, process_data as PD
    r = np.random.default_rng(3)
    n = 20000
    raw = np.stack([r.integers(0, 240, n), r.integers(0, 180, n), r.integers(0, 256, n), r.integers(0, 256, n),
                    r.integers(0, 256, n)], 1).astype(np.uint8).reshape(-1)
    ev = PD.decode_ncaltech101(raw)
    off = torch.tensor([0, n], dtype=torch.int64, device="cuda")
    img = D.rasterize(ev, off, 180, 240, False)[0].permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(img, E.event_arr_to_img(E.decode_ncaltech101(raw.tobytes()), 180, 240, False))
