"""Drop-in `models.EdgePredictor.EdgePredictor` on the MI355X HIP path.

Module tree and state_dict keys follow reference models/EdgePredictor.py:31-68
(including `spatial_proj`, which the reference builds but never uses — kept for
RNG-order and state_dict compatibility).  forward() accepts a
padded batch [B, V, vertex_dim]; `forward_ragged` runs samples with different vertex
counts in one pass, which is how PointCloudToWireframe replaces the reference's
serial per-sample loop."""
import torch
import torch.nn as nn

from wf3d import config
from wf3d.functional import EdgeFn, edge_index_lists


class EdgePredictor(nn.Module):
    def __init__(self, vertex_dim=3, hidden_dim=512, num_heads=8):
        super().__init__()
        if not 1 <= vertex_dim <= 8:
            raise ValueError("the HIP edge head takes 1..8 coordinates per vertex (reference default: 3)")
        if hidden_dim % 16 or hidden_dim % num_heads or (hidden_dim // num_heads) % 4:
            # said at construction, not by a kernel in the middle of a backward pass: the row kernels work on 16-byte
            # pieces of the hidden_dim / 4-wide layer, the attention kernels on 4-column pieces of a head
            raise ValueError(f"the HIP edge head needs hidden_dim % 16 == 0 and a head width (hidden_dim / num_heads) % 4 == 0 "
                             f"(got hidden_dim={hidden_dim}, num_heads={num_heads})")
        h = hidden_dim
        self.vertex_proj = nn.Sequential(
            nn.Linear(vertex_dim, h // 2), nn.LayerNorm(h // 2), nn.GELU(),
            nn.Linear(h // 2, h), nn.LayerNorm(h), nn.Dropout(0.1))
        self.attention = nn.MultiheadAttention(embed_dim=h, num_heads=num_heads, dropout=0.1, batch_first=True)
        self.spatial_proj = nn.Sequential(nn.Linear(vertex_dim, h // 4), nn.GELU(), nn.Linear(h // 4, h // 4))
        self.edge_mlp = nn.Sequential(
            nn.Linear(2 * h + 2 * vertex_dim + 1, h), nn.LayerNorm(h), nn.GELU(), nn.Dropout(0.1),
            nn.Linear(h, h // 2), nn.LayerNorm(h // 2), nn.GELU(), nn.Dropout(0.1),
            nn.Linear(h // 2, h // 4), nn.GELU(),
            nn.Linear(h // 4, 1))
        self._heads = num_heads

    def _get_edge_indices(self, num_vertices):
        """[E, 2] int64, all i<j in lexicographic order (reference :70-89)."""
        dev = next(self.parameters()).device
        if num_vertices < 2:
            return torch.tensor([], dtype=torch.long, device=dev)      # 1-D empty, as the reference yields
        return torch.triu_indices(num_vertices, num_vertices, 1, device=dev).t().contiguous()

    def _param_list(self):
        vp, at, em = self.vertex_proj, self.attention, self.edge_mlp
        return [vp[0].weight, vp[0].bias, vp[1].weight, vp[1].bias,
                vp[3].weight, vp[3].bias, vp[4].weight, vp[4].bias,
                at.in_proj_weight, at.in_proj_bias, at.out_proj.weight, at.out_proj.bias,
                em[0].weight, em[0].bias, em[1].weight, em[1].bias,
                em[4].weight, em[4].bias, em[5].weight, em[5].bias,
                em[8].weight, em[8].bias, em[10].weight, em[10].bias]

    def _dropout_ps(self):
        if not self.training:
            return (0.0, 0.0, 0.0, 0.0)
        return (self.vertex_proj[5].p, self.attention.dropout, self.edge_mlp[3].p, self.edge_mlp[7].p)

    def forward_ragged(self, vertices, counts):
        """vertices [B, Vmax, 3]; sample s uses its first counts[s] vertices.
        Returns zero-padded probabilities [B, max_s E_s]."""
        if len(counts) != vertices.shape[0]:
            raise ValueError(f"forward_ragged: {len(counts)} counts for a batch of {vertices.shape[0]}")
        counts = [min(int(c), vertices.shape[1]) for c in counts]      # `vertices[i, :count]` cannot take more than there is
        if any(c < 2 for c in counts):
            # the reference indexes a 1-D empty index tensor here (EdgePredictor.py:118)
            raise IndexError("too many indices for tensor of dimension 1")
        ps = self._dropout_ps()
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) if any(ps) else 0
        return EdgeFn.apply(vertices.float(), tuple(counts), self._heads, ps, seed, config.precision(),
                            *self._param_list())

    def forward(self, vertices):
        b, v, _ = vertices.shape
        probs = self.forward_ragged(vertices, [v] * b)
        return probs, edge_index_lists([v])[0]
