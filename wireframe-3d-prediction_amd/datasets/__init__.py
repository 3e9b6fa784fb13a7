"""Drop-in `datasets` package (reference datasets/__init__.py:12-18): build_dataset(cfg) -> {"train", "test"}."""
from .building3d import Building3DReconstructionDataset, DeviceCloudCache  # noqa: F401


def build_dataset(dataset_config):
    return {"train": Building3DReconstructionDataset(dataset_config, split_set="train"),
            "test": Building3DReconstructionDataset(dataset_config, split_set="test")}
