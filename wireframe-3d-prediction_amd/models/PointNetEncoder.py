"""Drop-in `models.PointNetEncoder.PointNetEncoder` on the MI355X HIP path.

Same constructor, attribute names and state_dict keys as the reference class
(reference models/PointNetEncoder.py:19-65: `mlp` Sequential with parameters at
indices 0,1,4,5,...,4n and `feature_fusion` at 0,1,3,4,6).  The nn modules here
are parameter containers only: forward never calls them, it hands their tensors
to wf3d.functional.EncoderFn (HIP kernels through the C ABI)."""
import torch.nn as nn

from wf3d import config
from wf3d.functional import EncoderFn, FusionFn


def _point_block(n_in, n_out):
    return [nn.Linear(n_in, n_out), nn.LayerNorm(n_out), nn.ReLU(inplace=True), nn.Dropout(0.0)]


class PointNetEncoder(nn.Module):
    def __init__(self, input_dim=8, hidden_dims=[512, 1024, 2048, 1024], output_dim=512):
        super().__init__()
        bad = [h for h in hidden_dims if h % 4 or not 0 < h <= 4096]
        if bad or output_dim % 2 or output_dim <= 0:
            # said here, not by a kernel in the middle of a backward pass: LayerNorm rows are handled in 16-byte pieces
            # (hidden widths, and the fusion MLP's 4 x / 2 x output_dim) and kept in registers up to 4096 columns
            raise ValueError(f"the HIP encoder needs hidden widths that are multiples of 4 (<= 4096) and an even output_dim "
                             f"(got hidden_dims={list(hidden_dims)}, output_dim={output_dim})")
        widths = [input_dim] + list(hidden_dims)
        stack = []
        for n_in, n_out in zip(widths[:-1], widths[1:]):
            stack += _point_block(n_in, n_out)
        stack.append(nn.Linear(widths[-1], output_dim))
        self.mlp = nn.Sequential(*stack)
        # declared (parameter-free) by the reference, never used in its forward
        self.global_max_pool = nn.AdaptiveMaxPool1d(1)
        self.global_avg_pool = nn.AdaptiveAvgPool1d(1)
        d = output_dim
        self.feature_fusion = nn.Sequential(
            nn.Linear(2 * d, 4 * d), nn.LayerNorm(4 * d), nn.ReLU(inplace=True),
            nn.Linear(4 * d, 2 * d), nn.LayerNorm(2 * d), nn.ReLU(inplace=True),
            nn.Linear(2 * d, d))
        self._n_hidden = len(hidden_dims)
        self.precision = None          # None -> wf3d.config.precision() ("bf16x3" | "fp32")

    def _mlp_params(self):
        ps = []
        for i in range(self._n_hidden):
            lin, ln = self.mlp[4 * i], self.mlp[4 * i + 1]
            ps += [lin.weight, lin.bias, ln.weight, ln.bias]
        last = self.mlp[4 * self._n_hidden]
        return ps + [last.weight, last.bias]

    def _fusion_params(self):
        ff = self.feature_fusion
        return [ff[0].weight, ff[0].bias, ff[1].weight, ff[1].bias,
                ff[3].weight, ff[3].bias, ff[4].weight, ff[4].bias,
                ff[6].weight, ff[6].bias]

    def encode(self, x):
        """(global, point_features, unmasked_pooled [B, 2C] = [mean | max]): the extra pools are what
        VertexPredictor would recompute from point_features (VertexPredictor.py:86-88)."""
        if x.dim() != 3:
            raise ValueError(f"expected (batch, num_points, input_dim), got {tuple(x.shape)}")
        pooled, pf, upooled = EncoderFn.apply(x.float(), self._n_hidden, self.precision or config.precision(),
                                              *self._mlp_params())
        g = FusionFn.apply(pooled, *self._fusion_params())
        return g, pf, upooled

    def forward(self, x):
        g, pf, _ = self.encode(x)
        return g, pf
