"""The split-GEMM kernels kept as selectable, measured baselines (WF3D_SPLIT_DMA = 2..7, WF3D_TN16 = 0/1: DESIGN.md §4)
must stay correct.  The selection is read once per process, so each variant runs tests/variant_check.py in a
child process (one at a time)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("env", [{"WF3D_SPLIT_DMA": "2"}, {"WF3D_SPLIT_DMA": "3"}, {"WF3D_SPLIT_DMA": "4"},
                                 {"WF3D_SPLIT_DMA": "5"}, {"WF3D_SPLIT_DMA": "6"}, {"WF3D_SPLIT_DMA": "7"}, {"WF3D_TN16": "0"},
                                 {"WF3D_SPLIT_DMA": "0"}])
def test_selectable_split_gemm_kernels(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "variant_check.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])
