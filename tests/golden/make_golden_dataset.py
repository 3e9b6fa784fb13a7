#!/usr/bin/env python3
"""Fixture generator for row f-3 (input pipeline): imports the REFERENCE's datasets.building3d from /root/reference (build
container only) and records what its Building3DReconstructionDataset returns for the three demo samples copied under
tests/golden/building3d/ (data files of the reference's demo dataset: train 100 (1276 points < 2560: sampled with
replacement), train 10040, test 10024), under fixed np.random seeds, with and without augmentation.
Output: tests/golden/dataset.npz (numeric only).   python tests/golden/make_golden_dataset.py"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_building3d", "/root/reference/datasets/building3d.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
root = os.path.join(HERE, "building3d")
case = 0
for split in ("train", "test"):
    for augment in (False, True):
        cfg = types.SimpleNamespace(root_dir=root, num_points=2560, use_color=True, use_intensity=True, normalize=True, augment=augment)
        ds = ref.Building3DReconstructionDataset(cfg, split_set=split)
        order = np.argsort([os.path.basename(f) for f in ds.pc_files])          # glob order is file-system dependent
        for rank, idx in enumerate(order):
            seed = 1000 + case
            np.random.seed(seed)
            item = ds[int(idx)]
            tag = f"c{case}"
            out[tag + ".meta"] = np.array([seed, int(augment), rank, 0 if split == "train" else 1])
            for k, v in item.items():
                out[tag + "." + k] = np.asarray(v)
            case += 1
# collate structure of a 2-sample batch
cfg = types.SimpleNamespace(root_dir=root, num_points=2560, use_color=True, use_intensity=True, normalize=True, augment=True)
ds = ref.Building3DReconstructionDataset(cfg, split_set="train")
order = np.argsort([os.path.basename(f) for f in ds.pc_files])
np.random.seed(77)
batch = ds.collate_batch([ds[int(i)] for i in order])
for k, v in batch.items():
    if isinstance(v, list):
        for j, t in enumerate(v):
            out[f"batch.{k}.{j}"] = t.numpy()
    else:
        out[f"batch.{k}"] = v.numpy()
out["ncases"] = np.array(case)
np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
print("wrote", case, "cases;", os.path.getsize(os.path.join(HERE, "dataset.npz")) // 1024, "KiB")
