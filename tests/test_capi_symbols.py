"""CPU-side checks of the drop-in boundary: libwf3d.so loads without a GPU and
exports every symbol include/wf3d.h declares, with the argument checks of the
C ABI reachable (no kernel is launched here)."""
import ctypes
import os
import re

import pytest

import helpers as H

HDR = os.path.join(H.ROOT, "include", "wf3d.h")


def declared_symbols():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(wf3d_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound():
    from wf3d import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"libwf3d.so does not export {n}"
        assert n in _lib.SIGNATURES, f"wf3d/_lib.py has no ctypes signature for {n}"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.wf3d_version() == 105


def test_library_exports_nothing_but_the_header():
    """exported is a subset of declared too: internal cross-file helpers (the LDS-DMA GEMM back end, the MFMA attention
    kernels, the error formatter) have hidden visibility."""
    import subprocess
    from wf3d import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if "wf3d" in ln.lower()}
    assert exported == set(declared_symbols()), sorted(exported ^ set(declared_symbols()))


def test_set_option_validates():
    from wf3d import _lib
    lib = _lib.load()
    assert lib.wf3d_set_option(b"tn_rounds", 2) == 0
    assert lib.wf3d_set_option(b"tn_rounds", 1) == 0
    assert lib.wf3d_set_option(b"tn_rounds", 9) == -1 and b"1..8" in lib.wf3d_last_error()
    assert lib.wf3d_set_option(b"gemm_cus", 224) == 0 and lib.wf3d_set_option(b"gemm_cus", 0) == 0
    assert lib.wf3d_set_option(b"gemm_cus", 3) == -1 and b"gemm_cus" in lib.wf3d_last_error()
    assert lib.wf3d_set_option(b"no_such_switch", 1) == -1 and b"unknown option" in lib.wf3d_last_error()
    assert lib.wf3d_set_option(None, 1) == -1


def test_argument_errors_reported_without_gpu():
    from wf3d import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.M, d.N, d.K = 4, 4, -1
    assert lib.wf3d_gemm(ctypes.byref(d), None) == -1
    assert b"negative" in lib.wf3d_last_error()
    d.K = 4                                  # null operands
    assert lib.wf3d_gemm(ctypes.byref(d), None) == -1
    assert lib.wf3d_row_stats(None, 4, 0, 0, 1e-5, None, None, None) == -1
    assert lib.wf3d_ln_act_bwd(None, None, 4, 6, None, None, None, None, 0, 0.0, 0, None, None, None, None, None, None, 0, None) == -4
    assert lib.wf3d_gemm_ws_bytes(32, 256, 4096, 0) > 0
    assert lib.wf3d_gemm_ws_bytes(4096, 4096, 64, 0) == 0


def test_struct_layout_matches_header():
    """GemmDesc field order/types must mirror wf3d_gemm_t."""
    from wf3d import _lib
    txt = open(HDR).read()
    body = re.search(r"typedef struct wf3d_gemm_t \{(.*?)\} wf3d_gemm_t;", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace("*", " ").split()
        # "int M, N, K" style declarations
        tail = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
        for nm in tail.split(","):
            fields.append(nm.replace("*", "").strip())
    assert fields == [f[0] for f in _lib.GemmDesc._fields_]


def test_ops_reject_cpu_tensors():
    import torch
    from wf3d import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.row_stats(torch.zeros(4, 8))
