"""Drop-in `models.VertexPredictor.VertexPredictor` on the MI355X HIP path.

Keeps the reference's module tree (models/VertexPredictor.py:27-61) including the
LAZILY created `point_pool_proj` (reference :94-97): it is built
on the first forward with default nn.Linear init from the global RNG, exactly
when and how the reference does, so optimizer/state_dict behaviour is identical."""
import torch.nn as nn

from wf3d.functional import UnmaskedPoolFn, VertexFn


def _stage(n_in, n_out):
    return nn.Sequential(nn.Linear(n_in, n_out), nn.LayerNorm(n_out), nn.ReLU(inplace=True), nn.Dropout(0.0))


class VertexPredictor(nn.Module):
    def __init__(self, global_feature_dim=512, max_vertices=64, vertex_dim=4):
        super().__init__()
        self.max_vertices = max_vertices
        self.vertex_dim = vertex_dim
        self.vertex_mlp1 = _stage(global_feature_dim, 4096)
        self.vertex_mlp2 = _stage(4096, 2048)
        self.vertex_mlp3 = _stage(2048, 2048)
        self.vertex_mlp4 = _stage(2048, 1024)
        self.final_layer = nn.Linear(1024, max_vertices * vertex_dim)
        self.residual_proj1 = nn.Linear(global_feature_dim, 2048)
        self.residual_proj2 = nn.Linear(global_feature_dim, 1024)

    def ensure_point_pool_proj(self, in_features, device):
        """Create the lazy layer if absent (same moment/shape/init as the reference)."""
        if not hasattr(self, "point_pool_proj"):
            self.point_pool_proj = nn.Linear(in_features, self.residual_proj1.in_features).to(device)
        return self.point_pool_proj

    def _param_list(self, with_pool):
        ps = []
        for st in (self.vertex_mlp1, self.vertex_mlp2, self.vertex_mlp3, self.vertex_mlp4):
            ps += [st[0].weight, st[0].bias, st[1].weight, st[1].bias]
        ps += [self.final_layer.weight, self.final_layer.bias,
               self.residual_proj1.weight, self.residual_proj1.bias,
               self.residual_proj2.weight, self.residual_proj2.bias]
        if with_pool:
            ps += [self.point_pool_proj.weight, self.point_pool_proj.bias]
        return ps

    def predict(self, global_features, upooled):
        """Vertex head from already-pooled point features, upooled = [mean | max] [B, 2C] (None -> no fusion)."""
        with_pool = upooled is not None
        if with_pool:
            self.ensure_point_pool_proj(upooled.shape[1], upooled.device)
        o, exist, counts = VertexFn.apply(global_features, upooled, self.max_vertices,
                                          self.vertex_dim, *self._param_list(with_pool))
        return {"vertices": o,                            # non-contiguous [B, V, 3] view of the [B, V, 4] output (:122)
                "existence_probabilities": exist,
                "actual_vertex_counts": counts}

    def forward(self, global_features, point_features, target_vertex_counts=None):
        if point_features is not None:
            return self.predict(global_features, UnmaskedPoolFn.apply(point_features.float()))
        return self.predict(global_features, None)
