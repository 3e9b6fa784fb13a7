"""Both split-GEMM tile kernels forced onto every shape (WF3D_SPLIT_DMA = 3: 256x128 / 32x32x16, 6: 256x256 / 16x16x32),
the 32x32x16 wgrad kernel (WF3D_TN16 = 0) and the register-staged fallback (WF3D_SPLIT_DMA = 0) must stay correct.  The selection is read once per process, so each variant runs tests/variant_check.py in a
child process (one at a time)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("env", [{"WF3D_SPLIT_DMA": "3"}, {"WF3D_SPLIT_DMA": "6"}, {"WF3D_TN16": "0"}, {"WF3D_SPLIT_DMA": "0"}])
def test_selectable_split_gemm_kernels(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "variant_check.py")], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])
