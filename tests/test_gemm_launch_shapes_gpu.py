"""The split GEMMs at the shapes the benchmark actually launches.

Every golden / frozen-gradient model case is small (<= 32 tiles) and therefore runs the few-tile kernels; the kernels
that carry 66 % of the step — the persistent 256x256 forward/dgrad kernel (selected at >= 2 x CUs full tiles) and the
XCD-mapped split-K wgrad kernel — are selected only at the encoder's real launch shapes.  Here those shapes run against
fp64 on EVERY output element (torch fp64 matmul on the GPU, row-chunked), once on random fp32 data (the hi*lo / lo*hi
products matter) and once on small-integer data (every fp32 sum exact: a slice dropped, doubled or mis-addressed shows
bit-exactly).  Reference lines replaced: models/PointNetEncoder.py:94 (the per-point Linear chain), its autograd
backward, and models/EdgePredictor.py:137 (edge MLP) at max_vertices = 256.
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers as H  # noqa: E402,F401  (sets sys.path)

TOL_SPLIT = 3e-5          # bf16x3 keeps ~16 significant bits per product; the gate on the path is 1e-4
M_ENC = 131072            # cfg2: 32 clouds x 4096 points
ROWS_EDGE = 1044480       # cfg5: 32 samples x 32,640 vertex pairs


@pytest.fixture(scope="module")
def ops():
    from wf3d import ops as o
    o._lib.load()
    return o


def dev():
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=dev()).manual_seed(seed)
    return torch.randn(*shape, generator=g, device=dev()) * scale


def ints(shape, mul, mod, off):
    n = 1
    for s in shape:
        n *= s
    return ((torch.arange(n, device=dev(), dtype=torch.int64) * mul) % mod - off).float().reshape(shape)


def worst_nt(got, A, B, bias, chunk=16384):
    """max |got - (A B^T + bias)| / max |ref| over all elements, fp64 reference computed in row chunks on the GPU."""
    Bd = B.double()
    bd = bias.double() if bias is not None else None
    err = torch.zeros((), dtype=torch.float64, device=dev())
    top = torch.zeros((), dtype=torch.float64, device=dev())
    for r in range(0, A.shape[0], chunk):
        want = A[r:r + chunk].double() @ Bd.T
        if bd is not None:
            want += bd
        err = torch.maximum(err, (got[r:r + chunk].double() - want).abs().max())
        top = torch.maximum(top, want.abs().max())
    return float(err / top), float(err)


def ref_tn(A, B, chunk=16384):
    """A^T B in fp64, accumulated over row chunks of the reduction index."""
    out = torch.zeros(A.shape[1], B.shape[1], dtype=torch.float64, device=dev())
    for r in range(0, A.shape[0], chunk):
        out += A[r:r + chunk].double().T @ B[r:r + chunk].double()
    return out


ENC_LAYERS = [(512, 1024), (1024, 2048), (2048, 1024), (1024, 512)]        # (K_in, N_out) of Linear 2..5


@pytest.mark.parametrize("K,N", ENC_LAYERS)
def test_forward_and_dgrad_at_encoder_launch_shapes(ops, K, N):
    """h·W^T + b (persistent kernel with the bias as accumulator start) and dz·W at M = 131,072."""
    M = M_ENC
    A, W, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.05), rnd(N, seed=3)
    As, Ws = ops.split_rows(A), ops.split_rows(W)
    got = ops.gemm_split(As, Ws, bias=bias)
    r, _ = worst_nt(got, A, W, bias)
    assert r < TOL_SPLIT, f"forward K={K} N={N}: {r:.2e}"
    got2 = ops.gemm_split(As, Ws)                                         # no-bias instantiation
    assert torch.equal(got2, ops.gemm_split(As, Ws, out=torch.full_like(got2, 7.0)))      # deterministic, overwrites `out`
    r, _ = worst_nt(got2, A, W, None)
    assert r < TOL_SPLIT
    del got, got2, As
    G = rnd(M, N, seed=4)                                                 # dgrad: dh = dz · W == gemm_split(dz_s, (W^T)_s)
    dh = ops.gemm_split(ops.split_rows(G), ops.split_rows(W, transpose=True))
    r, _ = worst_nt(dh, G, W.T.contiguous(), None)
    assert r < TOL_SPLIT, f"dgrad K={N} N={K}: {r:.2e}"


@pytest.mark.parametrize("K,N", ENC_LAYERS)
def test_forward_exact_on_integer_data_at_encoder_launch_shapes(ops, K, N):
    M = M_ENC
    A, W, bias = ints((M, K), 1, 11, 5), ints((N, K), 7, 13, 6), ints((N,), 1, 5, 2)
    got = ops.gemm_split(ops.split_rows(A), ops.split_rows(W), bias=bias)
    _, err = worst_nt(got, A, W, bias)
    assert err == 0.0


@pytest.mark.parametrize("Mo,No", [(1024, 512), (2048, 1024), (1024, 2048), (512, 1024)])
def test_wgrad_at_encoder_launch_shapes(ops, Mo, No):
    """dW = dz^T·h over K = 131,072 rows: XCD-mapped split-K of the 256x256 transposing-read kernel + slab fold."""
    K = M_ENC
    dz, h = rnd(K, Mo, seed=5), rnd(K, No, seed=6).relu_()
    dzs, hs = ops.split_rows(dz), ops.split_rows(h)
    assert ops.gemm_split_tn_ok(dzs, hs)
    got = ops.gemm_split_tn(dzs, hs)
    want = ref_tn(dz, h)
    r = float((got.double() - want).abs().max() / want.abs().max())
    assert r < TOL_SPLIT, f"wgrad {Mo}x{No}: {r:.2e}"
    assert torch.equal(got, ops.gemm_split_tn(dzs, hs))                   # fixed fold order: bit-identical on a rerun
    dzi, hi = ints((K, Mo), 1, 5, 2), ints((K, No), 3, 7, 3)
    goti = ops.gemm_split_tn(ops.split_rows(dzi), ops.split_rows(hi))
    assert torch.equal(goti.double(), ref_tn(dzi, hi))


@pytest.mark.parametrize("K,N", [(512, 256), (256, 128)])
def test_edge_mlp_gemms_at_cfg5_row_count(ops, K, N):
    """max_vertices = 256: 1,044,480 pair rows through Linear(512,256) / Linear(256,128) (EdgePredictor.py:61,65),
    forward, dgrad and the few-tile wgrad that gets a K range per CU."""
    R = ROWS_EDGE
    A, W, bias = rnd(R, K, seed=7), rnd(N, K, seed=8, scale=0.05), rnd(N, seed=9)
    As = ops.split_rows(A)
    got = ops.gemm_split(As, ops.split_rows(W), bias=bias)
    r, _ = worst_nt(got, A, W, bias)
    assert r < TOL_SPLIT, f"edge forward {K}->{N}: {r:.2e}"
    del got
    G = rnd(R, N, seed=10)
    Gs = ops.split_rows(G)
    dh = ops.gemm_split(Gs, ops.split_rows(W, transpose=True))
    r, _ = worst_nt(dh, G, W.T.contiguous(), None)
    assert r < TOL_SPLIT, f"edge dgrad {N}->{K}: {r:.2e}"
    del dh
    want = ref_tn(G, A)                                                   # dW [N, K]
    if ops.gemm_split_tn_ok(Gs, As):
        got = ops.gemm_split_tn(Gs, As)
    else:                                                                 # the transposed problem, as functional._wgrad_tn does
        got = ops.gemm_split_tn(As, Gs).t()
    r = float((got.double() - want).abs().max() / want.abs().max())
    assert r < TOL_SPLIT, f"edge wgrad {N}x{K}: {r:.2e}"


CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from wf3d import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(21)
M, K, N = 131072, 512, 1024
A = torch.randn(M, K, generator=g, device=dev); W = torch.randn(N, K, generator=g, device=dev) * 0.05; b = torch.randn(N, generator=g, device=dev)
got = ops.gemm_split(ops.split_rows(A), ops.split_rows(W), bias=b)
err = 0.0; top = 0.0
for r in range(0, M, 16384):
    want = A[r:r + 16384].double() @ W.double().T + b.double()
    err = max(err, float((got[r:r + 16384].double() - want).abs().max())); top = max(top, float(want.abs().max()))
print("REL", err / top)
"""


def test_persistent_kernel_with_reserved_cus_in_child_process():
    """WF3D_RESERVED_CUS=8 (read once per process): the persistent grid is 248 workgroups, every tile still computed once."""
    env = dict(os.environ, WF3D_RESERVED_CUS="8")
    out = subprocess.run([sys.executable, "-c", CHILD, H.PKG], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rel = float([ln for ln in out.stdout.splitlines() if ln.startswith("REL")][-1].split()[1])
    assert rel < TOL_SPLIT


@pytest.mark.parametrize("cus", [8, 200])
def test_persistent_kernel_on_fewer_workgroups(cus):
    """wf3d_set_option("gemm_cus", n): the claimed tiles are all computed, once each, whatever the grid."""
    from wf3d import _lib, ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(22)
    M, K, N = 131072, 512, 1024
    A = torch.randn(M, K, generator=g, device=dev)
    W = torch.randn(N, K, generator=g, device=dev) * 0.05
    As, Ws = ops.split_rows(A), ops.split_rows(W)
    want = ops.gemm_split(As, Ws)
    lib = _lib.load()
    assert lib.wf3d_set_option(b"gemm_cus", cus) == 0
    try:
        got = ops.gemm_split(As, Ws)
    finally:
        lib.wf3d_set_option(b"gemm_cus", 0)
    assert torch.equal(got, want)


@pytest.mark.parametrize("M,K,N", [(131072, 256, 256), (131072, 320, 256), (65536, 384, 512), (33024, 4096, 1024), (32768, 256, 4096),
                                   (131072, 192, 256), (16640, 2048, 2048)])
def test_persistent_kernel_at_the_edges_of_its_selection(ops, M, K, N):
    """Reductions of exactly 8 slices (the tile-claim protocol's minimum: draw at slice 0, consumed at slice 3, needed
    before the last two), odd multiples of 32 and fewer than 8 slices (the plain kernel takes those), a row count that
    leaves workgroups without a tile in the last round, a 16-column-tile panel, and just above the 2-tiles-per-CU
    threshold — every output element against fp64."""
    A, W, bias = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=0.05), rnd(N, seed=13)
    As, Ws = ops.split_rows(A), ops.split_rows(W)
    got = ops.gemm_split(As, Ws, bias=bias)
    r, _ = worst_nt(got, A, W, bias)
    assert r < TOL_SPLIT, f"M={M} K={K} N={N}: {r:.2e}"
    assert torch.equal(got, ops.gemm_split(As, Ws, bias=bias, out=torch.full_like(got, -3.0)))
