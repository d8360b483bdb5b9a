"""Portable counter-based tensor generator (TEST INFRASTRUCTURE ONLY).

Real-shape golden logits (tests/golden/real_*.npz) were produced by the reference
with weights/inputs from this generator, so the tests can rebuild the identical
61 M-parameter state dicts on the GPU box without shipping weight files
(SURVEY.md section 8c).  Pure numpy integer arithmetic: splitmix64 over
(seed, fnv1a64(name), element index) -- independent of torch's RNG/version.
"""
import math

import numpy as np

_M64 = (1 << 64) - 1


def _fnv1a64(name):
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & _M64
    return h


def _splitmix64(x):
    """x: uint64 ndarray -> uint64 ndarray (wrapping arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform01(name, n, seed, stream=0):
    base = (_fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & _M64) ^ ((stream * 0xD1B54A32D192ED03) & _M64)) & _M64
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(base)
    z = _splitmix64(_splitmix64(idx))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def portable_tensor(name, shape, lo=-1.0, hi=1.0, seed=0):
    """float32 ndarray of `shape`, uniform in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = _uniform01(name, n, seed)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def portable_input(shape, seed=0, name="input"):
    """float32 standard-normal-like input (Box-Muller on two portable uniform streams)."""
    n = int(np.prod(shape))
    u1 = _uniform01(name, n, seed, stream=1)
    u2 = _uniform01(name, n, seed, stream=2)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)
    return z.astype(np.float32).reshape(shape)


def portable_state_dict(shapes, seed=0):
    """Fill a {key: shape} map (a model's state_dict layout) with portable values.

    Scales follow the reference's default inits (U(+-1/sqrt(fan_in)) for Linear/Conv weights
    and biases) but every norm/affine/BN-statistics tensor is made NON-trivial so that the
    fused epilogues are actually exercised.  Returns {key: np.ndarray}.
    """
    out = {}
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        leaf = key.rsplit(".", 1)[-1]
        stem = key[: -len(leaf) - 1] if "." in key else ""
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, dtype=np.int64)
        elif leaf == "offset" and stem.rsplit(".", 1)[-1] in ("sfc_h", "sfc_w"):
            # CycleFC's registered buffer (cycle_mlp.py:95, 104-120) is structure, not a weight: its fixed integer pattern
            from .functional import cycle_offsets
            dy, dx = cycle_offsets(shape[1] // 2, (1, 3) if stem.endswith("sfc_h") else (3, 1))
            out[key] = np.array([v for pr in zip(dy, dx) for v in pr], dtype=np.float32).reshape(shape)
        elif leaf == "running_mean":
            out[key] = portable_tensor(key, shape, -0.1, 0.1, seed)
        elif leaf == "running_var":
            out[key] = portable_tensor(key, shape, 0.5, 1.5, seed)
        elif leaf in ("gamma_1", "gamma_2"):
            out[key] = portable_tensor(key, shape, 0.05, 0.2, seed)
        elif leaf == "alpha":
            out[key] = portable_tensor(key, shape, 0.9, 1.1, seed)
        elif leaf == "beta":
            out[key] = portable_tensor(key, shape, -0.1, 0.1, seed)
        elif leaf == "weight" and len(shape) >= 2:
            bound = 1.0 / math.sqrt(float(np.prod(shape[1:])))
            out[key] = portable_tensor(key, shape, -bound, bound, seed)
        elif leaf == "weight":                                   # norm scale
            out[key] = portable_tensor(key, shape, 0.9, 1.1, seed)
        elif leaf == "bias":
            wshape = shapes.get(stem + ".weight")
            if key.endswith("spatial_proj.bias"):                # g_mlp.py:15 inits this to 1.0
                out[key] = portable_tensor(key, shape, 0.9, 1.1, seed)
            elif wshape is not None and len(wshape) >= 2:
                bound = 1.0 / math.sqrt(float(np.prod(tuple(wshape)[1:])))
                out[key] = portable_tensor(key, shape, -bound, bound, seed)
            else:                                                # norm shift
                out[key] = portable_tensor(key, shape, -0.1, 0.1, seed)
        else:
            out[key] = portable_tensor(key, shape, -0.1, 0.1, seed)
    return out
