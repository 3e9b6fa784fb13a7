#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own model classes.

Run in the build container only (needs /root/reference, read-only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Nothing of the reference travels: the fixtures hold only numeric inputs'
recipes (names/seeds for oracle/detgen.py) and numeric outputs.  Every
parameter and input is regenerated from oracle/detgen.py, so the fixtures stay
small.  Dropout is zeroed in every case.

Cases (SURVEY.md §8c fixture plan):
  cfg1        B=1 N=1024 V=32 counts=[32], train mode, fwd + bwd summaries
  ragged      B=3 N=300 V=26 counts=[26,4,24], zero-padded points, one fully
              padded cloud, train mode, fwd + bwd summaries
  evalmode    B=2 N=200 V=12, eval mode (data-dependent counts)
  small_enc   PointNetEncoder(8,[32,64],16), full tensors fwd + bwd
  small_edge  EdgePredictor(3,64,2) V=7 and V=2, full tensors fwd + bwd
  small_vert  VertexPredictor(16, 5, 4), full tensors fwd + bwd
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("WF3D_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import detgen

from models.EdgePredictor import EdgePredictor              # noqa: E402  (reference)
from models.PointCloudToWireframe import PointCloudToWireframe  # noqa: E402
from models.PointNetEncoder import PointNetEncoder          # noqa: E402
from models.VertexPredictor import VertexPredictor          # noqa: E402

torch.set_num_threads(8)


def zero_dropout(m):
    for sub in m.modules():
        if isinstance(sub, torch.nn.Dropout):
            sub.p = 0.0
        if isinstance(sub, torch.nn.MultiheadAttention):
            sub.dropout = 0.0


def load_detgen(module, seed):
    arrs = detgen.fill_state_dict(module.state_dict(), seed)
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(torch.from_numpy(arrs[k]))


def grad_summary(named_params, out, prefix="grad."):
    for name, p in named_params:
        if p.grad is None:
            out[prefix + name + ".none"] = np.array(1, dtype=np.int8)
            continue
        g = p.grad.detach().double().reshape(-1)
        probe = torch.from_numpy(
            detgen.uniform("probe." + name, (g.numel(),), -1, 1, 7)).double()
        out[prefix + name + ".norm"] = np.array(g.norm().item())
        out[prefix + name + ".dot"] = np.array((g * probe).sum().item())
        out[prefix + name + ".head"] = g[:64].float().numpy()


def make_cloud(name, B, N, seed, pad_frac=0.0, dead_cloud=None):
    x = detgen.normalish(name, (B, N, 8), seed)
    if pad_frac > 0:
        u = detgen.uniform(name + ".pad", (B, N), 0, 1, seed)
        x[u < pad_frac] = 0.0
    if dead_cloud is not None:
        x[dead_cloud] = 0.0
    return x


def full_model_case(tag, B, N, V, counts, seed, train, pad_frac=0.0, dead_cloud=None):
    torch.manual_seed(0)
    model = PointCloudToWireframe(input_dim=8, max_vertices=V)
    zero_dropout(model)
    x = torch.from_numpy(make_cloud(tag + ".x", B, N, seed, pad_frac, dead_cloud))
    # first forward creates the lazy point_pool_proj (reference VertexPredictor.py:94-97)
    model.train()
    with torch.no_grad():
        model(x[:1, :8], torch.tensor([2]))
    load_detgen(model, seed)
    model.train(train)
    cnt = torch.tensor(counts, dtype=torch.long) if counts is not None else None
    out = model(x, cnt)
    res = {"meta.B": np.array(B), "meta.N": np.array(N), "meta.V": np.array(V),
           "meta.seed": np.array(seed), "meta.train": np.array(int(train)),
           "meta.pad_frac": np.array(pad_frac),
           "meta.dead_cloud": np.array(-1 if dead_cloud is None else dead_cloud),
           "meta.counts": np.array(counts if counts is not None else [], dtype=np.int64)}
    for k in ("vertices", "existence_probabilities", "edge_probs", "global_features"):
        res["out." + k] = out[k].detach().contiguous().numpy()
    res["out.actual_vertex_counts"] = out["actual_vertex_counts"].numpy()
    res["out.edge_index_lens"] = np.array([len(e) for e in out["edge_indices"]], dtype=np.int64)
    flat = [ij for e in out["edge_indices"] for ij in e]
    res["out.edge_indices_flat"] = np.array(flat, dtype=np.int64).reshape(-1, 2)
    # existence logit margin (Q10): how far sigmoid inputs are from the 0.5 flip
    pe = out["existence_probabilities"].detach()
    res["meta.exist_margin"] = np.array((pe - 0.5).abs().min().item())
    if train:
        cot = {k: torch.from_numpy(detgen.uniform(f"{tag}.cot.{k}", tuple(out[k].shape), -1, 1, seed))
               for k in ("vertices", "existence_probabilities", "edge_probs")}
        loss = sum((out[k] * cot[k]).sum() for k in cot)
        loss.backward()
        res["out.loss"] = np.array(loss.item())
        grad_summary(model.named_parameters(), res)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **res)
    print(tag, "edge_probs", tuple(out["edge_probs"].shape),
          "counts", out["actual_vertex_counts"].tolist(),
          "exist_margin", float(res["meta.exist_margin"]))


def small_encoder_case():
    tag = "small_enc"
    enc = PointNetEncoder(8, [32, 64], 16)
    load_detgen(enc, 3)
    x = torch.from_numpy(make_cloud(tag + ".x", 3, 37, 3, pad_frac=0.2, dead_cloud=2))
    g, pf = enc(x)
    cg = torch.from_numpy(detgen.uniform(tag + ".cot.g", tuple(g.shape), -1, 1, 3))
    cp = torch.from_numpy(detgen.uniform(tag + ".cot.pf", tuple(pf.shape), -1, 1, 3))
    ((g * cg).sum() + (pf * cp).sum()).backward()
    res = {"out.global": g.detach().numpy(), "out.point_features": pf.detach().numpy()}
    for n, p in enc.named_parameters():
        res["grad." + n] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **res)
    print(tag, "ok")


def small_edge_case():
    tag = "small_edge"
    ep = EdgePredictor(3, 64, 2)
    zero_dropout(ep)
    load_detgen(ep, 4)
    ep.train()
    res = {}
    for V in (7, 2):
        ep.zero_grad()
        v = torch.from_numpy(detgen.normalish(f"{tag}.v{V}", (2, V, 3), 4)).requires_grad_()
        probs, idx = ep(v)
        c = torch.from_numpy(detgen.uniform(f"{tag}.cot{V}", tuple(probs.shape), -1, 1, 4))
        (probs * c).sum().backward()
        res[f"V{V}.probs"] = probs.detach().numpy()
        res[f"V{V}.idx"] = np.array(idx, dtype=np.int64)
        res[f"V{V}.dverts"] = v.grad.numpy()
        for n, p in ep.named_parameters():
            if p.grad is not None:
                res[f"V{V}.grad." + n] = p.grad.numpy().copy()
    for V in (0, 1):
        try:
            ep(torch.zeros(1, V, 3))
            res[f"V{V}.raises"] = np.array(0)
        except IndexError:
            res[f"V{V}.raises"] = np.array(1)
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **res)
    print(tag, "ok; V<=1 raises:", int(res["V0.raises"]), int(res["V1.raises"]))


def small_vertex_case():
    tag = "small_vert"
    vp = VertexPredictor(16, 5, 4)
    g = torch.from_numpy(detgen.normalish(tag + ".g", (3, 16), 5)).requires_grad_()
    pf = torch.from_numpy(detgen.normalish(tag + ".pf", (3, 11, 16), 5)).requires_grad_()
    with torch.no_grad():
        vp(g, pf)                      # creates lazy layer
    load_detgen(vp, 5)
    out = vp(g, pf)
    cv = torch.from_numpy(detgen.uniform(tag + ".cot.v", tuple(out["vertices"].shape), -1, 1, 5))
    ce = torch.from_numpy(detgen.uniform(tag + ".cot.e", tuple(out["existence_probabilities"].shape), -1, 1, 5))
    ((out["vertices"] * cv).sum() + (out["existence_probabilities"] * ce).sum()).backward()
    res = {"out.vertices": out["vertices"].detach().contiguous().numpy(),
           "out.exist": out["existence_probabilities"].detach().numpy(),
           "out.counts": out["actual_vertex_counts"].numpy(),
           "grad.g": g.grad.numpy(), "grad.pf": pf.grad.numpy()}
    grad_summary(vp.named_parameters(), res)     # widths 4096/2048 are hard-coded: summaries only
    # point_features=None branch (VertexPredictor.py:101-102)
    out2 = vp(g.detach(), None)
    res["out.nopf.vertices"] = out2["vertices"].detach().contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, tag + ".npz"), **res)
    print(tag, "ok")


if __name__ == "__main__":
    small_encoder_case()
    small_edge_case()
    small_vertex_case()
    full_model_case("cfg1", 1, 1024, 32, [32], seed=11, train=True)
    full_model_case("ragged", 3, 300, 26, [26, 4, 24], seed=12, train=True,
                    pad_frac=0.1, dead_cloud=1)
    full_model_case("evalmode", 2, 200, 12, None, seed=13, train=False)
