"""Process-wide defaults of the MI355X build (not reference flags)."""
import os

from ._lib import BPM_BF16, BPM_F32

_PRECISION = os.environ.get("BPMULT_PRECISION", "bf16")


def set_precision(name: str) -> None:
    """'bf16' (throughput: bf16 MFMA operands, f32 accumulate / residual stream) or
    'f32' (parity: exact f32 MFMA end to end).  Read when a model first runs."""
    global _PRECISION
    if name not in ("bf16", "f32"):
        raise ValueError("precision must be 'bf16' or 'f32'")
    _PRECISION = name


def precision() -> str:
    return _PRECISION


def dtype_code(name: str) -> int:
    return BPM_BF16 if name == "bf16" else BPM_F32
