"""Pin the CPU oracle (oracle/reference_cpu.py) against the golden fixtures that
tests/golden/make_golden.py produced from the imported reference classes.

Tolerance: the oracle runs the same torch CPU kernels as the reference did, in
a different op grouping (functional calls, explicit attention), so agreement is
expected at a few fp32 ulps; 2e-6 relative to each tensor's scale is asserted.
"""
import numpy as np
import pytest
import torch

import helpers as H
from helpers import oracle, detgen

TOL = 2e-6


def test_small_encoder_full_tensors():
    gold = H.load_golden("small_enc")
    shapes = H.sub_shapes("encoder.", input_dim=8, hidden_dims=(32, 64), output_dim=16)
    P = oracle.params_from_numpy(detgen.fill_state_dict(shapes, 3))
    x = torch.from_numpy(H.make_cloud("small_enc.x", 3, 37, 3, pad_frac=0.2, dead_cloud=2))
    g, pf = oracle.encoder_forward(P, x, prefix="")
    assert H.rel_err(g.detach().numpy(), gold["out.global"]) < TOL
    assert H.rel_err(pf.detach().numpy(), gold["out.point_features"]) < TOL
    cg = torch.from_numpy(detgen.uniform("small_enc.cot.g", tuple(g.shape), -1, 1, 3))
    cp = torch.from_numpy(detgen.uniform("small_enc.cot.pf", tuple(pf.shape), -1, 1, 3))
    ((g * cg).sum() + (pf * cp).sum()).backward()
    for n, p in P.items():
        assert H.rel_err(p.grad.numpy(), gold["grad." + n]) < 5 * TOL, n


@pytest.mark.parametrize("V", [7, 2])
def test_small_edge_full_tensors(V):
    gold = H.load_golden("small_edge")
    shapes = H.sub_shapes("edge_predictor.", edge_hidden=64)
    P = oracle.params_from_numpy(detgen.fill_state_dict(shapes, 4))
    v = torch.from_numpy(detgen.normalish(f"small_edge.v{V}", (2, V, 3), 4)).requires_grad_()
    probs, idx = oracle.edge_forward(P, v, num_heads=2, prefix="")
    assert np.array_equal(np.array(idx, dtype=np.int64), gold[f"V{V}.idx"])   # bit-exact
    assert H.rel_err(probs.detach().numpy(), gold[f"V{V}.probs"]) < TOL
    c = torch.from_numpy(detgen.uniform(f"small_edge.cot{V}", tuple(probs.shape), -1, 1, 4))
    (probs * c).sum().backward()
    assert H.rel_err(v.grad.numpy(), gold[f"V{V}.dverts"]) < 5 * TOL
    for n, p in P.items():
        key = f"V{V}.grad.{n}"
        if key in gold:
            assert H.rel_err(p.grad.numpy(), gold[key]) < 5 * TOL, n
        else:                       # spatial_proj: constructed but unused
            assert p.grad is None, n


@pytest.mark.parametrize("V", [0, 1])
def test_edge_degenerate_vertex_count_raises(V):
    gold = H.load_golden("small_edge")
    assert int(gold[f"V{V}.raises"]) == 1
    shapes = H.sub_shapes("edge_predictor.", edge_hidden=64)
    P = oracle.params_from_numpy(detgen.fill_state_dict(shapes, 4), requires_grad=False)
    with pytest.raises(IndexError):
        oracle.edge_forward(P, torch.zeros(1, V, 3), num_heads=2, prefix="")


def test_small_vertex():
    gold = H.load_golden("small_vert")
    shapes = H.sub_shapes("vertex_predictor.", output_dim=16, max_vertices=5)
    P = oracle.params_from_numpy(detgen.fill_state_dict(shapes, 5))
    g = torch.from_numpy(detgen.normalish("small_vert.g", (3, 16), 5)).requires_grad_()
    pf = torch.from_numpy(detgen.normalish("small_vert.pf", (3, 11, 16), 5)).requires_grad_()
    out = oracle.vertex_forward(P, g, pf, 5, prefix="")
    assert H.rel_err(out["vertices"].detach().numpy(), gold["out.vertices"]) < TOL
    assert H.rel_err(out["existence_probabilities"].detach().numpy(), gold["out.exist"]) < TOL
    assert np.array_equal(out["actual_vertex_counts"].numpy(), gold["out.counts"])
    cv = torch.from_numpy(detgen.uniform("small_vert.cot.v", tuple(out["vertices"].shape), -1, 1, 5))
    ce = torch.from_numpy(detgen.uniform("small_vert.cot.e", tuple(out["existence_probabilities"].shape), -1, 1, 5))
    ((out["vertices"] * cv).sum() + (out["existence_probabilities"] * ce).sum()).backward()
    assert H.rel_err(g.grad.numpy(), gold["grad.g"]) < 5 * TOL
    assert H.rel_err(pf.grad.numpy(), gold["grad.pf"]) < 5 * TOL
    bad = H.check_grad_summaries(gold, [(n, p.grad) for n, p in P.items()], 5 * TOL)
    assert not bad, bad
    out2 = oracle.vertex_forward(P, g.detach(), None, 5, prefix="")
    assert H.rel_err(out2["vertices"].detach().numpy(), gold["out.nopf.vertices"]) < TOL


@pytest.mark.parametrize("tag", ["cfg1", "ragged", "evalmode"])
def test_full_model_cases(tag):
    gold = H.load_golden(tag)
    x, arrs, counts, V, seed, train = H.full_case_inputs(tag, gold)
    P = oracle.params_from_numpy(arrs, requires_grad=train)
    with torch.set_grad_enabled(train):
        out = oracle.model_forward(P, torch.from_numpy(x), counts, V, training=train)
    assert np.array_equal(out["actual_vertex_counts"].numpy(), gold["out.actual_vertex_counts"])
    lens = [len(e) for e in out["edge_indices"]]
    assert lens == gold["out.edge_index_lens"].tolist()
    flat = np.array([ij for e in out["edge_indices"] for ij in e], dtype=np.int64).reshape(-1, 2)
    assert np.array_equal(flat, gold["out.edge_indices_flat"])                   # bit-exact
    for k in ("vertices", "existence_probabilities", "edge_probs", "global_features"):
        assert H.rel_err(out[k].detach().numpy(), gold["out." + k]) < TOL, k
    if train:
        cot = H.full_case_cotangents(tag, out, seed)
        loss = sum((out[k] * cot[k]).sum() for k in cot)
        loss.backward()
        assert abs(loss.item() - float(gold["out.loss"])) < 1e-5 * max(1.0, abs(float(gold["out.loss"])))
        bad = H.check_grad_summaries(gold, [(n, p.grad) for n, p in P.items()], 1e-5)
        assert not bad, bad


def test_state_dict_shapes_cover_fixture_names():
    """Every parameter the reference exposed (fixture keys) is in the oracle's
    name->shape table and vice versa (80 tensors after the first forward)."""
    gold = H.load_golden("cfg1")
    names = {k[len("grad."):].rsplit(".", 1)[0] for k in gold if k.startswith("grad.")}
    table = set(oracle.state_dict_shapes(8, 32).keys())
    assert names == table
    assert len(table) == 80
