"""Closed-form deterministic tensor generator (TEST INFRASTRUCTURE ONLY).

Golden fixtures must not depend on torch's RNG stream (which could differ
between torch builds) nor carry 124 MB of weights, so every input cloud and
every parameter tensor of a fixture case is regenerated from this counter-based
generator: value[i] = f(splitmix64(key(name, seed) + i)).  numpy uint64
arithmetic only, so it gives the same bits on every machine.

Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module (it lives under oracle/).
"""
import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def uniform(name, shape, lo=-1.0, hi=1.0, seed=0):
    """float32 array of `shape`, uniform in [lo, hi), keyed by (name, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) << np.uint64(32)
    key = key ^ np.uint64((seed * 0x632BE5AB + 0x1234567) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + key
        bits = _splitmix64(ctr)
    # 24 high bits -> [0,1) exactly representable in float32
    u = (bits >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(name, shape, seed=0):
    """Roughly N(0,1) float32 (sum of 4 uniforms, variance-normalised)."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += uniform(f"{name}#n{k}", shape, -1.0, 1.0, seed).astype(np.float64)
    return (acc * np.sqrt(3.0 / 4.0)).astype(np.float32)


def fill_state_dict(sd, seed=0):
    """Return {name: np.float32 array} for every tensor of a state_dict-like
    mapping name -> tensor/array (only .shape is read).

    Rules (so LayerNorm affine terms and biases are all exercised):
      * LayerNorm-style weights (1-D, name ends '.weight')  : 1 + 0.2*u
      * 1-D '.bias' / 'in_proj_bias'                        : 0.1*u  (Linear) / 0.2*u (LN)
      * 2-D weights [out,in]                                : u / sqrt(in)
    """
    out = {}
    for name, t in sd.items():
        shape = tuple(getattr(t, "shape", t))
        if len(shape) == 2:
            out[name] = uniform(name, shape, -1.0, 1.0, seed) / np.float32(np.sqrt(shape[1]))
        elif name.endswith("weight"):
            out[name] = 1.0 + 0.2 * uniform(name, shape, -1.0, 1.0, seed)
        else:
            out[name] = 0.1 * uniform(name, shape, -1.0, 1.0, seed)
        out[name] = out[name].astype(np.float32)
    return out
