"""Thin, validated Python wrappers over the C ABI (include/bpmult_hip.h).

Everything here takes torch CUDA tensors only for their device pointers: shape,
dtype and contiguity are validated on the Python side (SURVEY.md 8(b) error
convention), the call is enqueued on torch's current stream, and a non-zero
return code becomes a RuntimeError.  No arithmetic happens in Python.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import (BPM_BF16, BPM_F32, F_ACCUM, F_ATOMIC, F_RELU, GEMM_NN, GEMM_NT, GEMM_TN, LN_OUT_F32, OUT_CT,
                   OUT_F32, OUT_HEADS, AttnProblem, GemmProblem)

__all__ = ["BPM_F32", "BPM_BF16", "GEMM_NT", "GEMM_NN", "GEMM_TN", "OUT_F32", "OUT_CT", "OUT_HEADS",
           "F_ACCUM", "F_RELU", "F_ATOMIC"]


def pad32(n: int) -> int:
    return (n + 31) // 32 * 32


def ct_torch(dtype: int) -> torch.dtype:
    return torch.bfloat16 if dtype == BPM_BF16 else torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need(t: torch.Tensor, dtype: torch.dtype, what: str) -> None:
    if not t.is_cuda:
        raise ValueError(f"{what}: expected a CUDA (HIP) tensor, the BPMulT hot path has no CPU fallback")
    if t.dtype != dtype:
        raise ValueError(f"{what}: expected {dtype}, got {t.dtype}")


# ----------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------
def gemm_problem(A: torch.Tensor, B: torch.Tensor, Cc: torch.Tensor, M: int, N: int, K: int,
                 lda: int, ldb: int, ldc: int, *, bias_n=None, bias_m=None, resid=None, ldr: int = 0,
                 gate=None, ldg: int = 0, gate_scale: float = 1.0, alpha: float = 1.0,
                 drop_p: float = 0.0, drop_seed: int = 0, drop_site: int = 0, flags: int = 0,
                 out_kind: int = OUT_F32, splitk: int = 1, heads=None) -> GemmProblem:
    """heads = (B, H, T, dh, dhp) for OUT_HEADS."""
    p = GemmProblem()
    p.A, p.B, p.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldb, p.ldc = lda, ldb, ldc
    p.bias_n, p.bias_m, p.resid, p.ldr = _p(bias_n), _p(bias_m), _p(resid), ldr
    p.gate, p.ldg, p.gate_scale = _p(gate), ldg, gate_scale
    p.alpha, p.drop_p, p.drop_seed, p.drop_site = alpha, drop_p, drop_seed, drop_site
    p.flags, p.out_kind, p.splitk = flags, out_kind, splitk
    if heads is not None:
        p.heads_B, p.heads_H, p.heads_T, p.heads_dh, p.heads_dhp = heads
    return p


def gemm_array(probs: Sequence[GemmProblem]):
    arr = (GemmProblem * len(probs))(*probs)
    return arr


def gemm_grouped(dtype: int, variant: int, probs, n: Optional[int] = None) -> None:
    """probs: a ctypes array from gemm_array() (cache it) or a list of GemmProblem."""
    if not isinstance(probs, C.Array):
        probs = gemm_array(probs)
    n = len(probs) if n is None else n
    for i in range(0, n, _lib.MAX_GROUP):
        k = min(_lib.MAX_GROUP, n - i)
        sub = C.cast(C.byref(probs, i * C.sizeof(GemmProblem)), C.POINTER(GemmProblem))
        _lib.check(_lib.lib().bpm_gemm_grouped(dtype, variant, sub, k, _stream()), "bpm_gemm_grouped")


# ----------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------
def attn_problem(Q, K, V, O, ldo, lse, B, H, T, S, dh, dhp, mask_off, *, dO=None, delta=None,
                 dQ=None, lddq=0, dK=None, lddk=0, dV=None, lddv=0, dq_scale=1.0,
                 drop_p=0.0, drop_seed=0, drop_site=0) -> AttnProblem:
    p = AttnProblem()
    p.Q, p.K, p.V, p.O, p.ldo, p.lse = _p(Q), _p(K), _p(V), _p(O), ldo, _p(lse)
    p.dO, p.delta = _p(dO), _p(delta)
    p.dQ, p.lddq, p.dK, p.lddk, p.dV, p.lddv = _p(dQ), lddq, _p(dK), lddk, _p(dV), lddv
    p.B, p.H, p.T, p.S, p.dh, p.dhp, p.mask_off = B, H, T, S, dh, dhp, mask_off
    p.dq_scale, p.drop_p, p.drop_seed, p.drop_site = dq_scale, drop_p, drop_seed, drop_site
    return p


def attn_array(probs: Sequence[AttnProblem]):
    return (AttnProblem * len(probs))(*probs)


def attn_fwd(dtype: int, probs) -> None:
    if not isinstance(probs, C.Array):
        probs = attn_array(probs)
    _lib.check(_lib.lib().bpm_attn_fwd(dtype, probs, len(probs), _stream()), "bpm_attn_fwd")


def attn_bwd(dtype: int, probs) -> None:
    if not isinstance(probs, C.Array):
        probs = attn_array(probs)
    _lib.check(_lib.lib().bpm_attn_bwd(dtype, probs, len(probs), _stream()), "bpm_attn_bwd")


# ----------------------------------------------------------------------------
# row kernels
# ----------------------------------------------------------------------------
def pack_rows_fwd(dtype, src, dst, B, T, Cn, ld, drop_p=0.0, seed=0, site=0) -> None:
    _need(src, torch.float32, "pack_rows_fwd.src")
    _need(dst, ct_torch(dtype), "pack_rows_fwd.dst")
    if not src.is_contiguous() or src.numel() != B * T * Cn or dst.numel() < T * B * ld:
        raise ValueError("pack_rows_fwd: bad shapes")
    _lib.check(_lib.lib().bpm_pack_rows_fwd(dtype, src.data_ptr(), dst.data_ptr(), B, T, Cn, ld, drop_p, seed, site, _stream()),
               "bpm_pack_rows_fwd")


def pack_rows_bwd(g, ldg, dsrc, B, T, Cn, drop_p=0.0, seed=0, site=0) -> None:
    _need(g, torch.float32, "pack_rows_bwd.g")
    _need(dsrc, torch.float32, "pack_rows_bwd.dsrc")
    _lib.check(_lib.lib().bpm_pack_rows_bwd(g.data_ptr(), ldg, dsrc.data_ptr(), B, T, Cn, drop_p, seed, site, _stream()),
               "bpm_pack_rows_bwd")


def pack_weights(dtype, table_dev: torch.Tensor, ndesc: int, total_blocks: int) -> None:
    _lib.check(_lib.lib().bpm_pack_weights(dtype, table_dev.data_ptr(), ndesc, total_blocks, _stream()), "bpm_pack_weights")


def embed_pos_fwd(x, table, out, T, B, d, scale, drop_p=0.0, seed=0, site=0) -> None:
    for t, n in ((x, "x"), (table, "table"), (out, "out")):
        _need(t, torch.float32, "embed_pos_fwd." + n)
    if not x.is_contiguous() or x.numel() != T * B * d or out.numel() != x.numel() or table.shape[1] != d:
        raise ValueError("embed_pos_fwd: bad shapes")
    _lib.check(_lib.lib().bpm_embed_pos_fwd(x.data_ptr(), table.data_ptr(), table.shape[0], out.data_ptr(), T, B, d,
                                           scale, drop_p, seed, site, _stream()), "bpm_embed_pos_fwd")


def embed_pos_bwd(dy, dx, T, B, d, scale, drop_p=0.0, seed=0, site=0, accumulate=False) -> None:
    _need(dy, torch.float32, "embed_pos_bwd.dy")
    _need(dx, torch.float32, "embed_pos_bwd.dx")
    _lib.check(_lib.lib().bpm_embed_pos_bwd(dy.data_ptr(), dx.data_ptr(), T, B, d, scale, drop_p, seed, site,
                                           int(accumulate), _stream()), "bpm_embed_pos_bwd")


def ln_fwd(out_dtype, x, gamma, beta, out, ldo, mean, rstd, R, d, eps=1e-5) -> None:
    _need(x, torch.float32, "ln_fwd.x")
    _lib.check(_lib.lib().bpm_ln_fwd(out_dtype, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ldo,
                                    mean.data_ptr(), rstd.data_ptr(), R, d, eps, _stream()), "bpm_ln_fwd")


def ln_bwd(dy, ldy, x, mean, rstd, gamma, add, dx, dgamma, dbeta, R, d) -> None:
    _need(dy, torch.float32, "ln_bwd.dy")
    _lib.check(_lib.lib().bpm_ln_bwd(dy.data_ptr(), ldy, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                    _p(add), dx.data_ptr(), _p(dgamma), _p(dbeta), R, d, _stream()), "bpm_ln_bwd")


def rows_cast(dtype, a, lda, R, Cn, *, b=None, ldb=0, dst_ct=None, ldd=0, dst_f32=None, ldf=0, colsum=None,
              drop_p=0.0, seed=0, site=0) -> None:
    _need(a, torch.float32, "rows_cast.a")
    _lib.check(_lib.lib().bpm_rows_cast(dtype, a.data_ptr(), lda, _p(b), ldb, _p(dst_ct), ldd, _p(dst_f32), ldf, _p(colsum),
                                       R, Cn, drop_p, seed, site, _stream()), "bpm_rows_cast")


def gmu2_fwd(a1, a2, ag, x1, x2, out, R, d) -> None:
    _lib.check(_lib.lib().bpm_gmu2_fwd(a1.data_ptr(), a2.data_ptr(), ag.data_ptr(), x1.data_ptr(), x2.data_ptr(),
                                      out.data_ptr(), R, d, _stream()), "bpm_gmu2_fwd")


def gmu2_bwd(dtype, dout, a1, a2, ag, x1, x2, da1, da2, dag, ldg, dx1, dx2, R, d) -> None:
    _lib.check(_lib.lib().bpm_gmu2_bwd(dtype, dout.data_ptr(), a1.data_ptr(), a2.data_ptr(), ag.data_ptr(), x1.data_ptr(),
                                      x2.data_ptr(), da1.data_ptr(), da2.data_ptr(), dag.data_ptr(), ldg,
                                      dx1.data_ptr(), dx2.data_ptr(), R, d, _stream()), "bpm_gmu2_bwd")
