"""wf3d — host side of the MI355X-native PointNet -> wireframe path.

`wf3d.ops`        one wrapper per C-ABI entry point (include/wf3d.h)
`wf3d.functional` torch.autograd.Function stages built from those ops
`wf3d.dist`       one-process-per-GPU data parallelism (RCCL gradient all-reduce)
"""
__all__ = ["ops"]
