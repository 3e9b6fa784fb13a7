"""Drop-in `models.PointCloudToWireframe.PointCloudToWireframe` (reference
models/PointCloudToWireframe.py:10-121) on the MI355X HIP path.

Differences in HOW, not WHAT: point_features is pooled once for both heads, and
the edge head runs all samples of the batch in one ragged pass instead of a
Python loop of batch-1 calls with a device->host sync each.  Outputs (keys,
shapes, dtypes, zero padding, edge_indices list-of-lists) are the reference's."""
import torch
import torch.nn as nn

from models.EdgePredictor import EdgePredictor
from models.PointNetEncoder import PointNetEncoder
from models.VertexPredictor import VertexPredictor
from wf3d.functional import edge_index_lists


class PointCloudToWireframe(nn.Module):
    def __init__(self, input_dim=8, max_vertices=64):
        super().__init__()
        self.max_vertices = max_vertices
        self.encoder = PointNetEncoder(input_dim=input_dim)
        self.vertex_predictor = VertexPredictor(global_feature_dim=512, max_vertices=max_vertices)
        self.edge_predictor = EdgePredictor(vertex_dim=3)
        self._count_cache = (None, -1, None)         # (the tensor object itself, its _version, host ints)

    def set_dropout(self, p):
        """Set every dropout probability of the edge head (the only non-zero ones)."""
        ep = self.edge_predictor
        ep.vertex_proj[5].p = ep.edge_mlp[3].p = ep.edge_mlp[7].p = float(p)
        ep.attention.dropout = float(p)
        return self

    def _host_counts(self, t):
        """Vertex counts as Python ints.  One device->host read per call (the reference does one
        `.item()` per sample, :80/:90).  The read is skipped only when the SAME tensor object is
        passed again unmodified, as train.py does every step (train.py:127): the cache holds a
        reference to that tensor — so its storage cannot be freed and handed to another batch's
        counts — and compares identity plus the autograd version counter, which every in-place
        torch op bumps.  (Writes that bypass the counter — `.data`, raw pointers — are not seen;
        pass a CPU tensor or a fresh tensor then.)  CPU tensors are read directly, no sync."""
        if not t.is_cuda:
            return [int(c) for c in t.tolist()]
        ref, ver, vals = self._count_cache
        if ref is t and ver == t._version:
            return vals
        vals = [int(c) for c in t.tolist()]
        self._count_cache = (t, t._version, vals)
        return vals

    def forward(self, point_cloud, target_vertex_counts=None):
        g, _pf, upooled = self.encoder.encode(point_cloud)
        vo = self.vertex_predictor.predict(g, upooled)
        verts = vo["vertices"]
        if self.training and target_vertex_counts is not None:
            counts = self._host_counts(target_vertex_counts)          # ground-truth counts (:77-86)
        else:
            counts = [int(c) for c in vo["actual_vertex_counts"].tolist()]   # data-dependent (:87-97)
        b = verts.shape[0]
        if len(counts) < b:
            # the reference reads target_vertex_counts[i] for every sample i of the batch (:78-80)
            raise IndexError(f"index {len(counts)} is out of bounds for dimension 0 with size {len(counts)}")
        counts = counts[:b]                                                   # entries beyond the batch are never read there
        v = self.max_vertices
        counts = [min(c, v) if c >= 0 else max(v + c, 0) for c in counts]    # what `[:count]` slicing does
        probs = self.edge_predictor.forward_ragged(verts, counts)
        return {"vertices": verts,
                "existence_probabilities": vo["existence_probabilities"],
                "edge_probs": probs,
                "edge_indices": edge_index_lists(counts),
                "global_features": g,
                "actual_vertex_counts": vo["actual_vertex_counts"]}
