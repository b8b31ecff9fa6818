"""Deterministic tensors shared by the golden-vector generator and the tests.

Weights and inputs are NOT stored in the fixtures: both sides regenerate them
from the tensor's name with numpy's PCG64 (bit-stable across platforms for a
given numpy major version), so a fixture only carries the reference's outputs.
"""
import zlib

import numpy as np


def det(name: str, shape, scale: float = 1.0, shift: float = 0.0) -> np.ndarray:
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return (rng.standard_normal(tuple(shape)) * scale + shift).astype(np.float32)


def det_param(name: str, shape) -> np.ndarray:
    """Parameter init used for parity runs.  Magnitudes follow the reference's
    initialisers (xavier / kaiming-uniform variance) but biases and LayerNorm
    affines are made non-trivial so every bias / affine path is exercised."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if "layer_norm" in name:                       # layer_norms.{i}.* and layer_norm.*
        return det(name, shape, 0.1, 1.0 if leaf == "weight" else 0.0)
    if leaf in ("bias", "in_proj_bias"):
        return det(name, shape, 0.05)
    fan_out, fan_in = shape[0], int(np.prod(shape[1:]))
    return det(name, shape, float(np.sqrt(2.0 / (fan_in + fan_out))))


def det_state_dict(shapes) -> dict:
    return {k: det_param(k, s) for k, s in shapes.items()}
