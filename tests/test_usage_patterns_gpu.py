"""Ways a training script may drive the drop-in model beyond `loss.backward()` on all outputs: several graphs alive, frozen
sub-modules, autocast regions, inference mode, backward from a subset of the outputs, copies, tensor hooks."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers as H  # noqa: E402,F401  (sys.path)


def _loss(o):
    return o["vertices"].sum() + o["existence_probabilities"].sum() + o["edge_probs"].sum()


@pytest.fixture(scope="module")
def setup():
    from models.PointCloudToWireframe import PointCloudToWireframe
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = PointCloudToWireframe(8, 6).to(dev)
    m.vertex_predictor.ensure_point_pool_proj(1024, dev)
    m.set_dropout(0.0)
    m.train()
    x = torch.randn(3, 300, 8, device=dev)
    c = torch.tensor([6, 2, 4], device=dev)
    m.zero_grad()
    _loss(m(x, c)).backward()
    ref = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    return m, x, c, ref


def _close(a, b, rtol=1e-4, atol=1e-6):
    return torch.allclose(a, b, rtol=rtol, atol=atol)


def test_two_graphs_alive_one_backward(setup):
    m, x, c, ref = setup
    m.zero_grad()
    a, b = m(x, c), m(x * 2, c)
    (_loss(a) + _loss(b)).backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_frozen_encoder_then_frozen_heads(setup):
    m, x, c, ref = setup
    for p in m.encoder.parameters():
        p.requires_grad_(False)
    try:
        m.zero_grad()
        _loss(m(x, c)).backward()
        assert all(p.grad is None for p in m.encoder.parameters())
        for n, p in m.named_parameters():
            if p.grad is not None:
                assert _close(p.grad, ref[n]), n
    finally:
        for p in m.parameters():
            p.requires_grad_(True)
    heads = list(m.vertex_predictor.parameters()) + list(m.edge_predictor.parameters())
    for p in heads:
        p.requires_grad_(False)
    try:
        m.zero_grad()
        _loss(m(x, c)).backward()
        for n, p in m.named_parameters():
            if n.startswith("encoder."):
                assert _close(p.grad, ref[n]), n
    finally:
        for p in m.parameters():
            p.requires_grad_(True)


def test_inside_an_autocast_region(setup):
    m, x, c, ref = setup
    with torch.autocast("cuda", dtype=torch.bfloat16):
        m.zero_grad()
        o = m(x, c)
        _loss(o).backward()
    assert o["vertices"].dtype == torch.float32            # the path computes in its own arithmetic, whatever autocast says
    for n, p in m.named_parameters():
        if p.grad is not None:
            assert _close(p.grad, ref[n], 1e-3, 1e-5), n


def test_inference_mode_and_no_grad(setup):
    m, x, c, ref = setup
    m.eval()
    try:
        with torch.inference_mode():
            o = m(x)
        assert o["edge_probs"].shape[0] == 3
    finally:
        m.train()
    with torch.no_grad():
        o = m(x, c)
    assert not o["vertices"].requires_grad


def test_backward_from_a_subset_of_the_outputs(setup):
    m, x, c, ref = setup
    m.zero_grad()
    m(x, c)["edge_probs"].sum().backward()
    assert m.vertex_predictor.final_layer.weight.grad is not None          # the edge head reads the predicted vertices
    m.zero_grad()
    m(x, c)["vertices"].sum().backward()
    assert m.edge_predictor.edge_mlp[0].weight.grad is None
    m.zero_grad()
    m(x, c)["global_features"].sum().backward()
    assert m.encoder.mlp[0].weight.grad is not None and m.vertex_predictor.final_layer.weight.grad is None


def test_deepcopy_state_dict_and_tensor_hooks(setup):
    m, x, c, ref = setup
    m2 = copy.deepcopy(m)
    m2.load_state_dict(m.state_dict())
    m2.zero_grad()
    _loss(m2(x, c)).backward()
    for n, p in m2.named_parameters():
        if p.grad is not None:
            assert _close(p.grad, ref[n]), n
    h = m.encoder.mlp[4].weight.register_hook(lambda g: g * 2)
    try:
        m.zero_grad()
        _loss(m(x, c)).backward()
    finally:
        h.remove()
    assert _close(m.encoder.mlp[4].weight.grad, 2 * ref["encoder.mlp.4.weight"])


def test_forward_and_backward_on_a_non_default_stream(setup):
    """Everything is queued on the caller's current stream (scratch buffers, the edge head's side-stream leaves and the
    persistent GEMM's claim counters are per stream): a step under another stream gives the same gradients."""
    m, x, c, ref = setup
    s = torch.cuda.Stream(x.device)
    s.wait_stream(torch.cuda.current_stream(x.device))
    with torch.cuda.stream(s):
        m.zero_grad()
        _loss(m(x, c)).backward()
    s.synchronize()
    for n, p in m.named_parameters():
        if p.grad is not None:
            assert _close(p.grad, ref[n]), n


def test_a_whole_step_is_capturable_into_a_graph(setup):
    """torch.cuda.graph (hipGraph) around forward + backward after a warm-up: no host read, no allocation outside the graph's
    pool, the persistent GEMM's counter reset and the edge head's side stream are captured with it — replays give the eager
    gradients bit for bit (2.9 -> 1.5 ms per step at cfg1's size, where the eager step is host-bound)."""
    m, x, c, ref = setup

    def step():
        for p in m.parameters():
            p.grad = None
        _loss(m(x, c)).backward()

    s = torch.cuda.Stream(x.device)
    s.wait_stream(torch.cuda.current_stream(x.device))
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream(x.device).wait_stream(s)
    torch.cuda.synchronize()
    eager = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    assert set(grads) == set(eager)
    for n in eager:
        assert torch.equal(grads[n], eager[n]), n
    for p in m.parameters():
        p.grad = None
