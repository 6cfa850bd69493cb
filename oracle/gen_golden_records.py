"""Generate tests/golden/records.npz by running the REFERENCE's own record decoder and loaders in the build container.

TEST INFRASTRUCTURE.  Needs /root/reference (absent on the GPU box -> exits).
  * process_data/process_dataset.py cannot be imported (configargparse, h5py): its ``ncaltech101`` FUNCTION is lifted by
    ``ast`` and executed unmodified -- directory walk, split file, the 5-byte loop of :48-63 and np.save included -- on
    a temporary input tree that holds seeded random .bin files.
  * mem/dataset_folder.py cannot be imported (torchvision / PIL datasets): ``dsec_npy_loader`` and ``imgnet_npy_loader``
    (:275-292) are lifted the same way and run on seeded .npy / .npz files.
For every vector: reference output == oracle restatement (oracle/events_np.py) asserted bit-exact, then inputs +
reference outputs are stored as the fixture.

    python -m oracle.gen_golden_records
"""
import ast
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _refimport as R            # noqa: E402
from oracle import events_np as E             # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def lift(path, names, ns):
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), os.path.basename(path), "exec"), ns)
    return ns


def main():
    if not R.available():
        sys.exit("no /root/reference here: fixtures are generated in the build container only")
    rng = np.random.default_rng(20260301)
    res = {}
    # ---- N-Caltech101: the reference's whole function on a temp tree
    ns = lift(R.REF + "/process_data/process_dataset.py", {"ncaltech101"}, {"np": np, "os": os, "print": lambda *a, **k: None})
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "in"), os.path.join(td, "out")
        os.makedirs(os.path.join(inp, "airplanes"))
        files = {"image_0001": 1500, "image_0002": 1, "image_0003": 0, "image_0004": 257}
        raws = {}
        for name, n in files.items():
            raw = rng.integers(0, 256, n * 5, dtype=np.uint8)
            if n >= 4:                                   # all-ones / all-zeros records and the extreme timestamps
                raw[:5] = 0xFF; raw[5:10] = 0x00; raw[10:15] = [1, 2, 0x7F, 0xFF, 0xFF]; raw[15:20] = [3, 4, 0x80, 0, 1]
            raws[name] = raw
            raw.tofile(os.path.join(inp, "airplanes", name + ".bin"))
        split = os.path.join(td, "split.txt")
        with open(split, "w") as f:
            # every file goes to "val": the reference builds train_set from a filter() iterator that val_set has already
            # exhausted (process_dataset.py:26-30), so with a split file only the val entries are ever processed
            for name in files:
                f.write(f"Caltech101/val/airplanes/{name}.bin\n")
        args = types.SimpleNamespace(input=inp, output=outp, split=split)
        ns["ncaltech101"]("airplanes", args)
        for i, name in enumerate(files):
            ref = np.load(os.path.join(outp, "val", "airplanes", name + ".npy"))
            ora = E.decode_ncaltech101(raws[name].tobytes())
            if ref.size == 0:
                ref = ref.reshape(0, 4)                  # np.array([]) of an empty file has shape (0,)
            assert ref.dtype == np.float64 and ref.shape == ora.shape and (ref == ora).all(), name
            res[f"ncaltech__{name}__raw"] = raws[name]
            res[f"ncaltech__{name}__events"] = ref
    # ---- loaders
    ns = lift(R.REF + "/mem/dataset_folder.py", {"dsec_npy_loader", "imgnet_npy_loader"}, {"np": np, "Any": object})
    with tempfile.TemporaryDirectory() as td:
        n = 5000
        cases = {
            "u16_i64_bool": (rng.integers(0, 640, n).astype(np.uint16), rng.integers(0, 480, n).astype(np.uint16),
                             np.sort(rng.integers(0, 2**40, n)).astype(np.int64), rng.integers(0, 2, n).astype(bool)),
            "i32_f64_u8": (rng.integers(0, 640, n).astype(np.int32), rng.integers(0, 480, n).astype(np.int32),
                           np.sort(rng.random(n) * 1e6), rng.integers(0, 2, n).astype(np.uint8)),
            # polarity values outside {0,1}: int8 wrap-around of `astype(np.int8) * 2 - 1`
            "i16_u32_i16wrap": (rng.integers(0, 640, n).astype(np.int16), rng.integers(0, 480, n).astype(np.int16),
                                np.sort(rng.integers(0, 2**31, n)).astype(np.uint32), rng.integers(-300, 300, n).astype(np.int16)),
        }
        for tag, (x, y, t, p) in cases.items():
            path = os.path.join(td, tag + ".npz")
            np.savez(path, x=x, y=y, t=t, p=p)
            ref = ns["imgnet_npy_loader"](path)
            ora = E.imgnet_struct_to_events(x, y, t, p)
            assert ref.dtype == np.float64 and ref.shape == (n, 4) and (ref == ora).all(), tag
            for k, v in zip("xytp", (x, y, t, p)):
                res[f"imgnet__{tag}__{k}"] = v
            res[f"imgnet__{tag}__events"] = ref
        # structured .npy (fields x, y, t, p): np.load gives a record array, data['x'] works the same way
        rec = np.zeros(777, dtype=[("x", "<u2"), ("y", "<u2"), ("t", "<i8"), ("p", "?")])
        rec["x"], rec["y"] = rng.integers(0, 640, 777), rng.integers(0, 480, 777)
        rec["t"], rec["p"] = np.sort(rng.integers(0, 10**9, 777)), rng.integers(0, 2, 777)
        path = os.path.join(td, "struct.npy")
        np.save(path, rec)
        ref = ns["imgnet_npy_loader"](path)
        assert (ref == E.imgnet_struct_to_events(rec["x"], rec["y"], rec["t"], rec["p"])).all()
        res["imgnet__struct__rec"] = rec
        res["imgnet__struct__events"] = ref
        for tag, dt in (("f64", np.float64), ("i64", np.int64), ("u16", np.uint16)):
            d = np.stack([rng.integers(0, 640, n), rng.integers(400, 480, n), np.sort(rng.integers(0, 60000, n)),
                          rng.integers(0, 2, n)], 1).astype(dt)
            path = os.path.join(td, f"dsec_{tag}.npy")
            np.save(path, d)
            ref = ns["dsec_npy_loader"](path)
            ora = E.dsec_to_events(d)
            assert ref.dtype == np.float64 and ref.shape == ora.shape and (ref == ora).all() and 0 < len(ref) < n
            res[f"dsec__{tag}__in"] = d
            res[f"dsec__{tag}__events"] = ref
    np.savez_compressed(os.path.join(OUT, "records.npz"), **res)
    print("records.npz:", len(res), "arrays,", os.path.getsize(os.path.join(OUT, "records.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
