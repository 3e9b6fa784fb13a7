"""Direct check of the CPU loss oracle (oracle/loss_cpu.py) against the reference's own losses/WireframeLoss.py at the shapes
the device-loss sweep uses (tests/test_loss.py::test_device_loss_matches_the_cpu_oracle_at_other_shapes) — values and input
gradients; runs only where /root/reference exists (the build container)."""
import os
import subprocess
import sys

import pytest

REF = os.environ.get("WF3D_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SCRIPT = r"""
import sys, importlib.util, io, contextlib
sys.dont_write_bytecode = True
sys.path.insert(0, %(root)r)
import torch
from oracle import loss_cpu
spec = importlib.util.spec_from_file_location("ref_loss", %(ref)r + "/losses/WireframeLoss.py")
ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
SHAPES = [(1, 1, [1], 1, 1), (1, 2, [0], 1, 1), (2, 3, [0, 0], 3, 3), (5, 4, [4, 1, 0, 2, 3], 6, 4), (64, 3, None, 3, 3),
          (3, 130, [130, 1, 77], 400, 8385), (2, 256, [256, 200], 32640, 32640), (7, 17, None, 136, 50)]
for B, V, counts, ep, et in SHAPES:
    g = torch.Generator().manual_seed(B * 1000 + V)
    if counts is None:
        counts = torch.randint(0, V + 1, (B,), generator=g).tolist()
    cnt = torch.tensor(counts, dtype=torch.long)
    pv, tv = torch.randn(B, V, 3, generator=g), torch.randn(B, V, 3, generator=g)
    pe = torch.sigmoid(2.0 * torch.randn(B, V, generator=g))
    pp = torch.sigmoid(2.0 * torch.randn(B, ep, generator=g))
    tl = (torch.rand(B, et, generator=g) > 0.7).float()
    te = (torch.arange(V)[None, :] < cnt[:, None]).float()
    tg = {"vertices": tv, "vertex_existence": te, "edge_labels": tl, "vertex_counts": cnt}
    res = []
    for fn in ("ref", "oracle"):
        p = {"vertices": pv.clone().requires_grad_(), "existence_probabilities": pe.clone().requires_grad_(), "edge_probs": pp.clone().requires_grad_()}
        with contextlib.redirect_stdout(io.StringIO()):
            out = ref.WireframeLoss(3.0, 1.5, 1.0)(p, tg) if fn == "ref" else loss_cpu.wireframe_loss(p, tg, 3.0, 1.5, 1.0)[0]
        out["total_loss"].backward()
        res.append((out, p))
    (a, pa), (b, pb) = res
    for k in ("total_loss", "vertex_loss", "existence_loss", "edge_loss"):
        assert abs(float(a[k]) - float(b[k])) <= 1e-6 * max(1.0, abs(float(a[k]))), ((B, V), k, float(a[k]), float(b[k]))
    for k in pa:
        ga, gb = pa[k].grad, pb[k].grad
        if ga is None or gb is None:
            assert (ga is None or float(ga.abs().max()) == 0.0) and (gb is None or float(gb.abs().max()) == 0.0), ((B, V), k)
        else:
            assert float((ga - gb).abs().max()) <= 1e-6 * max(float(ga.abs().max()), 1e-12), ((B, V), k)
print("LOSS_ORACLE_MATCHES_REFERENCE")
"""


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "losses", "WireframeLoss.py")), reason="reference not present")
def test_loss_oracle_equals_the_reference_at_the_sweep_shapes():
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"ref": REF, "root": ROOT}], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "LOSS_ORACLE_MATCHES_REFERENCE" in out.stdout, out.stderr[-3000:] + out.stdout[-1000:]
