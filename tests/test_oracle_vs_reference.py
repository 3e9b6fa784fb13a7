"""Direct oracle == reference check; runs only where /root/reference exists
(the build container).  On the GPU box the committed fixtures stand in."""
import os
import subprocess
import sys

import pytest

REF = os.environ.get("WF3D_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SCRIPT = r"""
import sys
sys.dont_write_bytecode = True
sys.path.insert(0, %(ref)r); sys.path.insert(0, %(root)r)
import torch
from models.PointCloudToWireframe import PointCloudToWireframe      # the reference
from oracle import reference_cpu as oracle
torch.manual_seed(5)
m = PointCloudToWireframe(8, 9)
for s in m.modules():
    if isinstance(s, torch.nn.Dropout): s.p = 0.0
    if isinstance(s, torch.nn.MultiheadAttention): s.dropout = 0.0
m.train()
x = torch.randn(2, 150, 8); x[0, 100:] = 0
cnt = torch.tensor([9, 3])
out = m(x, cnt)                                  # default torch init, lazy layer created here
P = oracle.params_from_module(m)
assert set(P) == set(oracle.state_dict_shapes(8, 9)), "state_dict key table drifted"
assert all(tuple(P[k].shape) == tuple(v) for k, v in oracle.state_dict_shapes(8, 9).items())
o2 = oracle.model_forward(P, x, cnt, 9, training=True)
assert out["edge_indices"] == o2["edge_indices"]
for k in ("vertices", "existence_probabilities", "edge_probs", "global_features"):
    d = (out[k] - o2[k]).abs().max().item(); s = out[k].abs().max().item()
    assert d <= 2e-6 * s, (k, d, s)
c = {k: torch.randn_like(out[k]) for k in ("vertices", "existence_probabilities", "edge_probs")}
sum((out[k] * c[k]).sum() for k in c).backward()
sum((o2[k] * c[k]).sum() for k in c).backward()
for n, p in m.named_parameters():
    if p.grad is None:
        assert P[n].grad is None, n
        continue
    d = (p.grad - P[n].grad).abs().max().item(); s = p.grad.abs().max().item()
    assert d <= 1e-5 * s + 1e-12, (n, d, s)
print("ORACLE_MATCHES_REFERENCE")
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference not present")
def test_oracle_matches_imported_reference():
    # separate interpreter: the reference's `models` namespace must not shadow the product's
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"ref": REF, "root": ROOT}],
                       capture_output=True, text=True, timeout=600,
                       env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert "ORACLE_MATCHES_REFERENCE" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
