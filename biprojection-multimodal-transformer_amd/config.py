"""Process-wide defaults of the MI355X build (not reference flags)."""
import os

from ._lib import BPM_BF16, BPM_F32

PRECISIONS = ("bf16", "f32", "bf16x3")
_PRECISION = os.environ.get("BPMULT_PRECISION", "bf16")


def set_precision(name: str) -> None:
    """'bf16' (throughput: bf16 MFMA operands, f32 accumulate / residual stream), 'f32' (parity: exact f32 MFMA end to
    end) or 'bf16x3' (parity-grade and faster than f32: everything as in f32 mode except the large linear-layer products,
    which run as three bf16 MFMA products of split operands -- ~2^-16 relative per product).  Read when a model first runs."""
    global _PRECISION
    if name not in PRECISIONS:
        raise ValueError("precision must be 'bf16', 'f32' or 'bf16x3'")
    _PRECISION = name


def precision() -> str:
    return _PRECISION


def dtype_code(name: str) -> int:
    """Storage / compute type of the activations and weight shadows (bf16x3 keeps fp32 storage)."""
    if name not in PRECISIONS:
        raise ValueError(f"unknown precision {name!r}")
    return BPM_BF16 if name == "bf16" else BPM_F32


def is_x3(name: str) -> bool:
    return name == "bf16x3"
