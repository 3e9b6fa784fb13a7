"""Row f-4 — evaluation post-processing (reference evaluate.py:74-110, eval/ap_calculator.py) — against fixtures recorded
from the REFERENCE's APCalculator / hausdorff_distance_line on synthetic predictions around a real demo wireframe
(tests/golden/make_golden_eval.py -> tests/golden/eval.npz).

CPU: the drop-in eval.ap_calculator must reproduce the Hausdorff matrix, the helper functions and every accumulated counter /
final metric.  GPU: the device Hausdorff kernel and the batched thresholding + end-point kernel must give the same numbers
through wf3d.postprocess.evaluate_batch."""
import contextlib
import io

import numpy as np
import pytest
import torch

import helpers as H

COUNTERS = ("tp_corners", "tp_fp_corners", "tp_fn_corners", "distance", "tp_edges", "wed", "tp_fp_edges", "tp_fn_edges")
FINAL = ("average_corner_offset", "average_wed", "corners_precision", "corners_recall", "corners_f1", "edges_precision",
         "edges_recall", "edges_f1")


def _z_first(vertices, edges):
    if len(edges) == 0:
        return np.empty((0, 2, 3))
    ev = np.stack((vertices[edges[:, 0]], vertices[edges[:, 1]]), axis=1)
    return ev[np.arange(len(ev))[:, None], np.flip(np.argsort(ev[:, :, -1]), axis=1)]


def _check_counters(calc, want, c):
    got = np.array([float(calc.ap_dict[k]) for k in COUNTERS])
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9), (c, got, want)


def test_ap_calculator_matches_reference_fixture():
    from eval import ap_calculator as ap
    g = H.load_golden("eval")
    hd = ap.hausdorff_distance_line(g["hd.p"].copy(), g["hd.t"].copy())
    assert hd.shape == g["hd.matrix"].shape and np.allclose(hd, g["hd.matrix"], rtol=1e-12, atol=1e-12)
    assert ap.hausdorff_distance_line(np.empty((0, 2, 3)), g["hd.t"]).size == 0
    assert np.array_equal(ap.remove_corners(g["rc.a"].copy(), g["rc.b"].copy()), g["rc.out"])
    assert np.array_equal(ap.computer_edges(g["ce.edges"], g["ce.verts"]), g["ce.out"])
    calc = ap.APCalculator(distance_thresh=1)
    gt_v, gt_e, pairs = g["gt_v"], g["gt_e"], g["pairs"]
    for c in range(3):
        pv, pr = g[f"c{c}.vertices"].copy(), g[f"c{c}.probs"]
        pd_edges = pairs[pr > 0.5]
        calc.compute_metrics({"predicted_vertices": pv[None], "predicted_edges": pd_edges[None],
                              "pred_edges_vertices": _z_first(pv, pd_edges).reshape(1, -1, 2, 3),
                              "wf_vertices": gt_v.copy()[None], "wf_edges": gt_e.copy()[None],
                              "wf_edges_vertices": _z_first(gt_v, gt_e).reshape(1, -1, 2, 3)})
        _check_counters(calc, g[f"c{c}.counters"], c)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        calc.output_accuracy()
    assert "Edges F1" in buf.getvalue()
    got = np.array([float(calc.ap_dict[k]) for k in FINAL])
    assert np.allclose(got, g["final"], rtol=1e-9, atol=1e-12), (got, g["final"])
    calc.reset()
    assert calc.ap_dict["tp_edges"] == 0 and "corners_f1" in calc.ap_dict


@pytest.mark.gpu
def test_device_postprocessing_matches_reference_fixture():
    from eval import ap_calculator as ap
    from wf3d import postprocess
    from wf3d.functional import edge_index_lists
    dev = torch.device("cuda:0")
    g = H.load_golden("eval")
    hd = postprocess.hausdorff_lines(g["hd.p"], g["hd.t"], 20, dev)
    assert np.allclose(hd, g["hd.matrix"], rtol=1e-12, atol=1e-12)
    gen = np.random.RandomState(1)
    big_p, big_t = gen.normal(0, 3, (500, 2, 3)).astype(np.float32), gen.normal(0, 3, (40, 2, 3)).astype(np.float32)
    assert np.allclose(postprocess.hausdorff_lines(big_p, big_t, 20, dev), ap.hausdorff_distance_line(big_p, big_t), rtol=1e-12, atol=1e-12)
    # the three fixture samples as ONE batch of model-style predictions (vertices is a strided view, as the model returns it)
    V = int(g["V"])
    verts4 = torch.zeros(3, V, 4)
    verts4[:, :, :3] = torch.from_numpy(np.stack([g[f"c{c}.vertices"] for c in range(3)]))
    pred = {"vertices": verts4.to(dev)[:, :, :3],
            "edge_probs": torch.from_numpy(np.stack([g[f"c{c}.probs"] for c in range(3)])).to(dev),
            "edge_indices": edge_index_lists([V, V, V])}
    assert np.array_equal(np.array(pred["edge_indices"][0]), g["pairs"])
    ws = postprocess.wireframes_from_predictions(pred)
    for c, w in enumerate(ws):
        pd_edges = g["pairs"][g[f"c{c}.probs"] > 0.5]
        assert np.array_equal(w["pd_edges"], pd_edges)
        assert np.array_equal(w["pred_vertices"], g[f"c{c}.vertices"])
        assert np.array_equal(w["pd_edges_vertices"], _z_first(g[f"c{c}.vertices"], pd_edges).astype(np.float32).reshape(-1, 2, 3))
    ap.set_device(dev)
    try:
        calc = ap.APCalculator(distance_thresh=1)
        gt_v, gt_e = torch.from_numpy(g["gt_v"]), torch.from_numpy(g["gt_e"].astype(np.float32))
        for c in range(3):
            one = {"vertices": pred["vertices"][c:c + 1], "edge_probs": pred["edge_probs"][c:c + 1], "edge_indices": pred["edge_indices"][c:c + 1]}
            postprocess.evaluate_batch(one, [gt_v], [gt_e], calc)
            _check_counters(calc, g[f"c{c}.counters"], c)
        # ragged batch: a sample with fewer vertices than max_vertices keeps its own (shorter) edge list
        pred2 = {"vertices": pred["vertices"][:2], "edge_probs": pred["edge_probs"][:2].clone(), "edge_indices": edge_index_lists([V, 5])}
        pred2["edge_probs"][1, 10:] = 0.0
        w2 = postprocess.wireframes_from_predictions(pred2)
        assert len(pred2["edge_indices"][1]) == 10 and (w2[1]["pd_edges"] < 5).all()
        assert np.array_equal(w2[1]["pd_edges"], np.array(pred2["edge_indices"][1])[g["c1.probs"][:10] > 0.5])
    finally:
        ap.set_device(None)
