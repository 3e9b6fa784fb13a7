"""Decision capture for the decision-frozen gradient check (TEST INFRASTRUCTURE, like the rest of oracle/: used by
tests/test_frozen_grad_gpu.py and __graft_entry__.smoke()).

A ReLU / arg-max network's gradient is piecewise constant in its pre-activations; to compare a HIP backward with an fp64
oracle element by element, the oracle is run with the HIP forward's own 0/1 decisions (oracle.reference_cpu.model_forward
(frozen=...)).  This module reads those decisions off the autograd nodes of a PointCloudToWireframe output dict."""


def _sx8_positive(h_s):
    """ReLU decision of the forward kernel itself, read off the sx8 operand it wrote: h > 0  <=>  its bf16 high part > 0
    (bf16 keeps fp32's exponent range, so no positive value rounds to zero)."""
    import torch
    R, C = h_s.shape
    planes = h_s.contiguous().view(torch.bfloat16).view(R, C // 8, 16)
    return (planes[:, :, :8] > 0).reshape(R, C).cpu()


def capture_decisions(out, model, border=5e-7, kernel_masks=False):
    """Read the piecewise-constant decisions the HIP forward took (ReLU masks of every LayerNorm+ReLU,
    pool arg-max rows) off the autograd nodes of a PointCloudToWireframe output dict, BEFORE backward.

    Returns (frozen dict for oracle.model_forward(frozen=...), n_border): n_border counts activations
    whose LayerNorm output lies within `border` of 0 — there the mask recomputed here could differ from
    the one the kernels take (they evaluate the same fp32 expression in their own order), so a
    decision-frozen comparison is only meaningful on inputs with n_border == 0.

    kernel_masks=True: for the per-point layers whose activated output exists as an sx8 operand (bf16x3 mode), the mask
    is read from that operand instead — the forward kernel's own decision, exact whatever the margin — and those layers
    do not count towards n_border.  (Batches of thousands of points always contain activations within 5e-7 of 0.)"""
    relu, n_border = {}, 0

    def mask(name, z, mu, rs):
        nonlocal n_border
        sd = model.state_dict()
        g, b = sd[name + ".weight"], sd[name + ".bias"]
        v = ((z.double() - mu.double()[:, None]) * rs.double()[:, None]) * g.double() + b.double()
        n_border += int((v.abs() < border).sum())
        relu[name] = (v > 0).cpu()

    vfn = out["existence_probabilities"].grad_fn
    ffn = out["global_features"].grad_fn
    assert type(vfn).__name__.startswith("VertexFn") and type(ffn).__name__.startswith("FusionFn")
    efn = ffn.next_functions[0][0]
    assert type(efn).__name__.startswith("EncoderFn")
    x2, valid, zs, stats, hs, arg_m, arg_u, cnt = efn.saved
    for i, (z, (mu, rs)) in enumerate(zip(zs, stats)):
        if kernel_masks and hs[i] is not None:
            relu[f"encoder.mlp.{4 * i + 1}"] = _sx8_positive(hs[i])
        else:
            mask(f"encoder.mlp.{4 * i + 1}", z, mu, rs)
    pooled, f0, s0, f3, s3 = ffn.saved
    mask("encoder.feature_fusion.1", f0, *s0)
    mask("encoder.feature_fusion.4", f3, *s3)
    pooled_v, e, z1, s1, z2, s2, z3, s3v, c, z4, s4, d = vfn.saved
    for k, (z, s) in enumerate(((z1, s1), (z2, s2), (z3, s3v), (z4, s4)), start=1):
        mask(f"vertex_predictor.vertex_mlp{k}.1", z, *s)
    frozen = {"relu": relu,
              "argmax": {"enc_masked": arg_m.long().cpu(), "vert_unmasked": arg_u.long().cpu()}}
    return frozen, n_border
