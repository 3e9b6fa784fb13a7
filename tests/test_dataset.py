"""Row f-3 — the input pipeline (reference datasets/building3d.py:95-190) — against fixtures recorded from the REFERENCE's
own dataset class on three files of its demo data (tests/golden/make_golden_dataset.py -> tests/golden/dataset.npz).

CPU: the drop-in Building3DReconstructionDataset must return the reference's dict (keys, dtypes, shapes, values) for the same
np.random seed — the sampled rows, flips and rotation included — and collate like it.
GPU: DeviceCloudCache.batch() must produce the same batch as collate_batch([dataset[i] ...]) from resident clouds."""
import os
import types

import numpy as np
import pytest
import torch

import helpers as H

ROOT = os.path.join(H.GOLDEN, "building3d")


def _cfg(augment, **kw):
    d = dict(root_dir=ROOT, num_points=2560, use_color=True, use_intensity=True, normalize=True, augment=augment)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _sorted(ds):
    return np.argsort([os.path.basename(f) for f in ds.pc_files])


def test_items_match_reference_fixtures():
    from datasets import build_dataset
    g = H.load_golden("dataset")
    n = int(g["ncases"])
    assert n == 6
    for c in range(n):
        seed, augment, rank, split = [int(v) for v in g[f"c{c}.meta"]]
        ds = build_dataset(_cfg(bool(augment)))["train" if split == 0 else "test"]
        idx = int(_sorted(ds)[rank])
        np.random.seed(seed)
        item = ds[idx]
        want = {k[len(f"c{c}."):]: v for k, v in g.items() if k.startswith(f"c{c}.") and not k.endswith(".meta")}
        assert set(item) == set(want), (set(item) ^ set(want))
        for k, w in want.items():
            got = np.asarray(item[k])
            assert got.dtype == w.dtype and got.shape == w.shape, (c, k, got.dtype, w.dtype, got.shape, w.shape)
            if got.dtype.kind == "f":
                assert np.allclose(got, w, rtol=0, atol=1e-7 * max(1.0, float(np.abs(w).max()))), (c, k, np.abs(got - w).max())
            else:
                assert np.array_equal(got, w), (c, k)
        assert item["point_clouds"].shape == (2560, 8) and item["point_clouds"].dtype == np.float32
        assert np.abs(item["point_clouds"][:, :3]).max() <= 1.0 + 1e-6           # max-norm normalised


def test_collate_matches_reference_fixture():
    from datasets import Building3DReconstructionDataset
    g = H.load_golden("dataset")
    ds = Building3DReconstructionDataset(_cfg(True), split_set="train")
    np.random.seed(77)
    batch = ds.collate_batch([ds[int(i)] for i in _sorted(ds)])
    keys = {k.split(".")[1] for k in g if k.startswith("batch.")}
    assert set(batch) == keys
    for k in keys:
        if isinstance(batch[k], list):
            for j, t in enumerate(batch[k]):
                w = g[f"batch.{k}.{j}"]
                assert t.dtype == torch.float32 and tuple(t.shape) == w.shape
                assert np.allclose(t.numpy(), w, rtol=0, atol=1e-6), (k, j)
        else:
            w = g[f"batch.{k}"]
            assert tuple(batch[k].shape) == w.shape and batch[k].numpy().dtype == w.dtype, k
            assert np.allclose(batch[k].numpy(), w, rtol=0, atol=1e-7 * max(1.0, float(np.abs(w).max()))), k


def test_wireframe_io_and_column_modes(tmp_path):
    from datasets.building3d import Building3DReconstructionDataset, load_wireframe, save_wireframe
    v, e = load_wireframe(os.path.join(ROOT, "train", "wireframe", "100.obj"))
    assert v.dtype == np.float64 and v.shape[1] == 3 and e.shape[1] == 2 and (e[:, 0] <= e[:, 1]).all()
    p = str(tmp_path / "w.obj")
    save_wireframe(v, e, p)
    v2, e2 = load_wireframe(p)
    assert np.allclose(v, v2) and {tuple(x) for x in e} == {tuple(x) for x in e2}
    ds = Building3DReconstructionDataset(_cfg(False, use_color=False, use_intensity=False, num_points=100), "test")
    assert ds[0]["point_clouds"].shape == (100, 3)
    ds = Building3DReconstructionDataset(_cfg(False, use_color=True, use_intensity=False, num_points=0), "test")
    pc = ds[0]["point_clouds"]
    assert pc.shape[1] == 7 and 0 <= pc[:, 3:].min() and pc[:, 3:].max() < 1.0
    ds = Building3DReconstructionDataset(_cfg(False, use_color=False, use_intensity=True), "test")
    with pytest.raises(ValueError):                      # the reference's own concatenate fails for this combination (:107)
        ds[0]


@pytest.mark.gpu
@pytest.mark.parametrize("augment", [False, True])
def test_device_cache_batches_equal_host_batches(augment):
    from datasets import Building3DReconstructionDataset, DeviceCloudCache
    dev = torch.device("cuda:0")
    ds = Building3DReconstructionDataset(_cfg(augment), split_set="train")
    cache = DeviceCloudCache(ds, dev)
    for seed, idxs in ((5, [0, 1]), (6, [1, 1, 0])):
        np.random.seed(seed)
        want = ds.collate_batch([ds[i] for i in idxs])
        np.random.seed(seed)
        got = cache.batch(idxs)
        assert set(got) == set(want)
        assert got["point_clouds"].is_cuda and got["point_clouds"].dtype == torch.float32
        assert tuple(got["point_clouds"].shape) == (len(idxs), 2560, 8)
        assert float((got["point_clouds"].cpu() - want["point_clouds"]).abs().max()) < 1e-6
        for k in want:
            if k == "point_clouds":
                continue
            if isinstance(want[k], list):
                for a, b in zip(got[k], want[k]):
                    assert a.shape == b.shape and float((a - b).abs().max()) < 1e-6, k
            else:
                assert got[k].shape == want[k].shape and np.allclose(got[k].numpy(), want[k].numpy(), rtol=0, atol=1e-6), k     # centroid / max_distance: fp64 sums of ~6.5e6 m in another order
    # the model takes the batch as is
    from models.PointCloudToWireframe import PointCloudToWireframe
    model = PointCloudToWireframe(8, 16).to(dev)
    model.train()
    counts = torch.tensor([min(16, len(v)) for v in got["wf_vertices"]])
    out = model(got["point_clouds"], counts)
    assert torch.isfinite(out["vertices"]).all()


def test_xyz_parser_takes_the_files_own_columns_comments_and_rejects_ragged_rows(tmp_path):
    """np.loadtxt semantics of the C parser: the column count is the file's own (3-, 4-, 7-column .xyz files
    load), `#` comments are skipped, rows of different lengths are an error rather than a silent reshape."""
    from datasets.building3d import _read_xyz
    p = tmp_path / "a.xyz"
    p.write_text("# header line\n1.5 2 3\n4 5e0 6   # trailing comment\n\n-7 8 9.25\n")
    got = _read_xyz(str(p))
    assert got.shape == (3, 3) and np.array_equal(got, np.loadtxt(str(p), dtype=np.float64))
    q = tmp_path / "b.xyz"
    q.write_text("1 2 3 4 5 6 7\n8 9 10 11 12 13 14\n")
    assert _read_xyz(str(q)).shape == (2, 7)
    with pytest.raises(ValueError, match="expected 8"):
        _read_xyz(str(q), ncols=8)
    r = tmp_path / "c.xyz"
    r.write_text("1 2 3 4\n5 6 7 8\n9 10 11 12\n13 14 15 16\n1 2 3 4 5 6 7 8\n")      # 24 values: a multiple of 8, but ragged
    with pytest.raises(ValueError, match="different numbers of columns"):
        _read_xyz(str(r))
    with pytest.raises(OSError):
        _read_xyz(str(tmp_path / "missing.xyz"))
