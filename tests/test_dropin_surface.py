"""CPU checks of the drop-in Python surface: class names,
signatures, state_dict keys/shapes, lazy parameter, edge index order, and that
nothing computes on the CPU (no fallback)."""
import inspect
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H
from helpers import oracle


def test_signatures_and_state_dict_match_reference_table():
    from models.EdgePredictor import EdgePredictor
    from models.PointCloudToWireframe import PointCloudToWireframe
    from models.PointNetEncoder import PointNetEncoder
    from models.VertexPredictor import VertexPredictor
    assert list(inspect.signature(PointCloudToWireframe.__init__).parameters) == ["self", "input_dim", "max_vertices"]
    assert list(inspect.signature(PointCloudToWireframe.forward).parameters) == ["self", "point_cloud", "target_vertex_counts"]
    assert list(inspect.signature(PointNetEncoder.__init__).parameters) == ["self", "input_dim", "hidden_dims", "output_dim"]
    assert list(inspect.signature(VertexPredictor.__init__).parameters) == ["self", "global_feature_dim", "max_vertices", "vertex_dim"]
    assert list(inspect.signature(VertexPredictor.forward).parameters) == ["self", "global_features", "point_features", "target_vertex_counts"]
    assert list(inspect.signature(EdgePredictor.__init__).parameters) == ["self", "vertex_dim", "hidden_dim", "num_heads"]
    m = PointCloudToWireframe(input_dim=8, max_vertices=26)
    assert m.max_vertices == 26
    tab = oracle.state_dict_shapes(8, 26, with_lazy=False)
    sd = m.state_dict()
    assert set(sd) == set(tab) and all(tuple(sd[k].shape) == tuple(tab[k]) for k in tab)
    assert sum(p.numel() for p in m.parameters()) == 30373097          # train.py:43-45 prints this for V=26
    m.vertex_predictor.ensure_point_pool_proj(1024, "cpu")
    assert set(m.state_dict()) == set(oracle.state_dict_shapes(8, 26))
    assert sum(p.numel() for p in m.parameters()) == 30897897
    # evaluate.py:51-52 recovers V from final_layer
    assert m.state_dict()["vertex_predictor.final_layer.weight"].shape[0] // 4 == 26


def test_edge_indices_order_is_the_reference_loop():
    from models.EdgePredictor import EdgePredictor
    from wf3d.functional import edge_index_lists
    ep = EdgePredictor()
    for v in (2, 3, 7, 64):
        want = oracle.edge_index_pairs(v)
        assert ep._get_edge_indices(v).tolist() == want
        assert edge_index_lists([v])[0] == want
    assert ep._get_edge_indices(1).dim() == 1 and ep._get_edge_indices(0).numel() == 0


def test_no_cpu_fallback():
    from models.PointCloudToWireframe import PointCloudToWireframe
    m = PointCloudToWireframe(8, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 16, 8), torch.tensor([2]))


REF = os.environ.get("WF3D_REFERENCE", "/root/reference")
INIT_SCRIPT = r"""
import sys, importlib.util, types
sys.dont_write_bytecode = True
import torch
def load_pkg(root, alias):
    # import <root>/models/*.py under a private package name so both trees coexist
    pkg = types.ModuleType(alias); pkg.__path__ = [root + "/models"]; sys.modules[alias] = pkg
    mods = {}
    for n in ("EdgePredictor", "PointNetEncoder", "VertexPredictor", "PointCloudToWireframe"):
        src = open(f"{root}/models/{n}.py").read().replace("from models.", f"from {alias}.")
        m = types.ModuleType(f"{alias}.{n}"); sys.modules[f"{alias}.{n}"] = m
        exec(compile(src, f"{root}/models/{n}.py", "exec"), m.__dict__)
        mods[n] = m
    return mods["PointCloudToWireframe"].PointCloudToWireframe
sys.path.insert(0, %(pkg)r)
Ref = load_pkg(%(ref)r, "refmodels")
Ours = load_pkg(%(pkg)r, "ourmodels")
torch.manual_seed(1234); a = Ref(8, 12)
torch.manual_seed(1234); b = Ours(8, 12)
sa, sb = a.state_dict(), b.state_dict()
assert list(sa) == list(sb), "state_dict key order differs"
for k in sa:
    assert torch.equal(sa[k], sb[k]), k
assert [n for n, _ in a.named_parameters()] == [n for n, _ in b.named_parameters()]
print("INIT_IDENTICAL")
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference not present")
def test_same_seed_gives_reference_initialisation():
    r = subprocess.run([sys.executable, "-c", INIT_SCRIPT % {"ref": REF, "pkg": H.PKG}],
                       capture_output=True, text=True, timeout=600,
                       env={**os.environ, "PYTHONDONTWRITEBYTECODE": "1"})
    assert "INIT_IDENTICAL" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_edge_index_lists_are_read_only_and_equal_to_plain_lists():
    from wf3d.functional import edge_index_lists
    a = edge_index_lists([4, 2])
    assert a[0] == [[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]] and a[1] == [[0, 1]]
    assert isinstance(a[0], list) and isinstance(a[0][0], list)
    assert np.array(a[0]).shape == (6, 2)
    with pytest.raises(TypeError):
        a[0].pop()
    with pytest.raises(TypeError):
        a[0][0][1] = 9
    assert edge_index_lists([4])[0] == oracle.edge_index_pairs(4)
