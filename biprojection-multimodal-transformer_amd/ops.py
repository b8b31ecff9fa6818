"""Thin Python wrappers over the C ABI (include/bpmult_hip.h).

Torch CUDA tensors are used only for their device pointers.  Problem structs
are plain ctypes structures built once per (layer, op) and replayed every step
with a fresh dropout seed; the call is enqueued on torch's current stream and a
non-zero return code becomes a RuntimeError.  No arithmetic happens in Python,
and there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (BPM_BF16, BPM_BF16X3, BPM_F32, F_ACCUM, F_ATOMIC, F_KPAD, F_RELU, GEMM_NN, GEMM_NT, GEMM_TN, GEMM_MAX_GROUP, MAX_GROUP, OUT_CT,
                   OUT_F32, OUT_HEADS, AttnProblem, CastProblem, EmbedProblem, GemmProblem, GmuProblem,
                   LnProblem, PackProblem)


def pad32(n: int) -> int:
    return (n + 31) // 32 * 32


def ct_torch(dtype: int) -> torch.dtype:
    return torch.bfloat16 if dtype == BPM_BF16 else torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class DeviceSeed(int):
    """A dropout seed the kernels read from device memory when they run (captured graphs): BPM_SEED_INDIRECT | address
    of a uint64.  Only values of this type may have bit 63 set; a plain int with that bit is rejected below instead of
    being dereferenced as a pointer on the device."""

    def __new__(cls, tensor: torch.Tensor):
        if tensor.dtype != torch.int64 or tensor.numel() < 1 or not (tensor.is_cuda or _DRY_RUN):
            raise ValueError("DeviceSeed: a CUDA int64 tensor holding the seed")
        return super().__new__(cls, _lib.SEED_INDIRECT | tensor.data_ptr())


def _seed(seed: int) -> int:
    if (seed & _lib.SEED_INDIRECT) and not isinstance(seed, DeviceSeed):
        raise ValueError("dropout seeds passed by value are 63-bit (bit 63 marks a device-resident seed: ops.DeviceSeed)")
    return int(seed)


# tests/test_plan_tables_cpu.py only: lets the launch TABLES (host logic: pointers, shapes, flags) be built from host
# tensors without a GPU.  Nothing can be launched from them -- every bpm_* entry point needs device pointers.
_DRY_RUN = False


def _p(t) -> Optional[int]:
    if t is None or isinstance(t, int):     # raw device address (e.g. an offset into a flat buffer)
        return t
    if not t.is_cuda and not _DRY_RUN:
        raise ValueError("expected a CUDA (HIP) tensor: the BPMulT hot path has no CPU fallback")
    return t.data_ptr()


def _f32(t, what):
    if t is not None and not isinstance(t, int) and t.dtype != torch.float32:
        raise ValueError(f"{what}: expected float32, got {t.dtype}")
    return _p(t)


def array(cls, probs: Sequence):
    """ctypes array of problem structs (cache it; it is replayed every step)."""
    return (cls * len(probs))(*probs)


def _chunks(arr, cls, n, limit=MAX_GROUP):
    n = len(arr) if n is None else n
    for i in range(0, n, limit):
        k = min(limit, n - i)
        yield C.cast(C.byref(arr, i * C.sizeof(cls)), C.POINTER(cls)), k


def _as_array(cls, probs):
    return probs if isinstance(probs, C.Array) else array(cls, probs)


# ----------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------
def gemm_problem(A, B, Cc, M: int, N: int, K: int, lda: int, ldb: int, ldc: int, *, bias_n=None, bias_m=None,
                 resid=None, ldr: int = 0, gate=None, ldg: int = 0, gate_scale: float = 1.0, alpha: float = 1.0,
                 drop_p: float = 0.0, drop_site: int = 0, colsum=None, flags: int = 0, out_kind: int = OUT_F32,
                 splitk: int = 1, heads=None, colsum_a=None, batch=None) -> GemmProblem:
    """A, B, C: tensors or raw device addresses (int).  heads = (B, H, T, dh, dhp) for OUT_HEADS.
    batch = (n, stride_a, stride_b, stride_c): n products of this shape, operands / output of element i start i * stride
    elements further on (BPM_GEMM_BATCHED)."""
    p = GemmProblem()
    p.A, p.B, p.C = (x if isinstance(x, int) else _p(x) for x in (A, B, Cc))
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldb, p.ldc = lda, ldb, ldc
    p.bias_n, p.bias_m = _f32(bias_n, "bias_n"), _f32(bias_m, "bias_m")
    p.resid, p.ldr = _f32(resid, "resid"), ldr
    p.gate, p.ldg, p.gate_scale = _p(gate), ldg, gate_scale
    p.alpha, p.drop_p, p.drop_site = alpha, drop_p, drop_site
    p.colsum = colsum if isinstance(colsum, int) else _f32(colsum, "colsum")
    p.flags, p.out_kind, p.splitk = flags, out_kind, splitk
    if heads is not None:
        p.heads_B, p.heads_H, p.heads_T, p.heads_dh, p.heads_dhp = heads
    p.colsum_a = colsum_a if isinstance(colsum_a, int) else _f32(colsum_a, "colsum_a")
    if batch is not None:
        p.batch, p.batch_stride_a, p.batch_stride_b, p.batch_stride_c = batch
        p.flags |= _lib.F_BATCHED
    return p


def gemm_grouped(dtype: int, variant: int, probs, seed: int = 0, n: Optional[int] = None, x3: bool = False) -> None:
    """x3 (or an array tagged `.x3 = True` where the launch tables are built): the parity-grade fast mode -- fp32 operands,
    each product as three bf16 MFMA products of split operands (see _X3Plan); launches the LDS-DMA kernel cannot take
    run as exact fp32 products."""
    arr = _as_array(GemmProblem, probs)
    L, s = _lib.lib(), _stream()
    if (x3 or getattr(arr, "x3", False)) and dtype == BPM_F32 and variant in _X3_VARIANTS:
        plan = getattr(arr, "_x3_plan", None)
        if plan is None or plan.n != n:
            plan = _X3Plan(arr, variant, n)
            arr._x3_plan = plan              # tables are replayed every step; lists were turned into a fresh array above
        if plan.ok:
            plan.run(L, variant, _seed(seed), s)
            if plan.rest is None:
                return
            arr, n = plan.rest, None            # the problems the split path does not take: exact fp32 products
    for sub, k in _chunks(arr, GemmProblem, n, GEMM_MAX_GROUP):
        _lib.check(L.bpm_gemm_grouped(dtype, variant, sub, k, _seed(seed), s), "bpm_gemm_grouped")


# ---- bf16x3: fp32 products as three bf16 MFMA products of split operands -------------------------------------------
# x = hi + lo (hi = bf16(x), lo = bf16(x - hi)); x y ~ hi hi + hi lo + lo hi with f32 accumulation: ~2^-16 relative per
# product (the lo lo term and the rounding of lo), against 2^-9 for plain bf16 operands.  Every operand of an eligible
# launch is split by bpm_split_rows into a [rows, hi plane | lo plane] bf16 image (cached per operand view: the buffers
# of the hot path are static) right before the launch; weight shadows -- operands inside a registered static range -- are
# split again only after the shadows were refreshed.
# lab knob: BPMULT_X3_ONLY=nt,nn restricts the split path to those operand arrangements (bisecting a parity failure)
_X3_VARIANTS = {{"nt": GEMM_NT, "nn": GEMM_NN, "tn": GEMM_TN}[v] for v in __import__("os").environ.get("BPMULT_X3_ONLY", "nt,nn,tn").split(",") if v}
_X3_BUFFERS = {}                # (ptr, rows, cols, ld) -> split image
_X3_STATIC = []                 # [lo, hi) address ranges of operands that only change at a refresh (weight shadows)
_X3_STATIC_SPLITS = {}          # (ptr, rows, cols, ld) -> SplitProblem of every static operand seen so far


def x3_drop_static(lo: int, hi: int) -> None:
    """A parameter store is gone (its model was moved or deleted): forget its static operands -- their addresses are."""
    if (lo, hi) in _X3_STATIC:
        _X3_STATIC.remove((lo, hi))
    for k in [k for k in _X3_STATIC_SPLITS if lo <= k[0] < hi]:
        del _X3_STATIC_SPLITS[k]


def x3_register_static(lo: int, hi: int) -> None:
    if (lo, hi) not in _X3_STATIC:
        _X3_STATIC.append((lo, hi))


def _split(L, probs, s) -> None:
    arr = _as_array(_lib.SplitProblem, probs)
    for sub, k in _chunks(arr, _lib.SplitProblem, None):
        _lib.check(L.bpm_split_rows(sub, k, s), "bpm_split_rows")


def x3_refresh_static() -> None:
    """Split every static operand again (ParamStore.refresh_shadows, after the weight shadows changed).  A host-side
    decision, outside any captured graph -- like the shadow refresh itself; the launch plans never split static operands
    except the first time they meet one."""
    if _X3_STATIC_SPLITS:
        _split(_lib.lib(), list(_X3_STATIC_SPLITS.values()), _stream())


_X3_EPOCH = [0]                 # bumped once per forward pass (x3_new_step)
_X3_FRESH = {}                  # operand key -> epoch of its last split (host program order)


def x3_new_step() -> None:
    _X3_EPOCH[0] += 1


class _X3Plan:
    """Split problems + the BPM_BF16X3 problem table of one grouped GEMM launch (built once per launch table).
    Problems the split path does not take (fewer than 256 rows / columns / k, row bias, ...) stay behind in `rest` and run
    as exact fp32 products.  Operands listed in the table's `x3_presplit` attribute (set where the launch tables are
    built: forward activations a backward launch reads again, gradients an earlier launch of the same layer has split)
    are not split again when their image is from this step."""

    def __init__(self, arr, variant: int, n: Optional[int]):
        self.n = n
        cnt = len(arr) if n is None else n
        P = [arr[i] for i in range(cnt)]
        xk, yk = variant != GEMM_TN, variant == GEMM_NT
        take = [p for p in P if self._eligible(p)]
        if take and variant == GEMM_TN:               # one 256 x 256 tile per CU must roughly fill the chip (as in bf16 mode)
            if sum(((p.M + 255) // 256) * ((p.N + 255) // 256) for p in take) * 8 < 256 * 3:
                take = []
        self.ok = bool(take)
        self.rest = None
        self._keys = []
        if not self.ok:
            return
        if len(take) < cnt:
            ids = {C.addressof(p) for p in take}
            self.rest = array(GemmProblem, [p for p in P if C.addressof(p) not in ids])
        presplit = getattr(arr, "x3_presplit", ())
        dyn, new_static, out = [], [], []
        seen = {}
        for p in take:
            q = GemmProblem()
            C.memmove(C.byref(q), C.byref(p), C.sizeof(GemmProblem))
            for side, rows, cols, ld in (("A", p.M if xk else p.K, p.K if xk else p.M, p.lda),
                                         ("B", p.N if yk else p.K, p.K if yk else p.N, p.ldb)):
                ptr = getattr(p, side)
                key = (ptr, rows, cols, ld)
                ldp = (cols + 127) // 128 * 128
                buf = _X3_BUFFERS.get(key)
                if buf is None:
                    # torch.empty, NOT zeros: bpm_split_rows writes every column of both planes (pads included), and an
                    # initialising kernel on the allocating stream would race with the other stream's split of the same
                    # operand (main runs ahead of the side stream: its split + product would be wiped by the late fill)
                    buf = _X3_BUFFERS[key] = torch.empty(rows, 2 * ldp, device="cuda", dtype=torch.bfloat16)
                if key not in seen:
                    seen[key] = True
                    sp = _lib.SplitProblem()
                    sp.src, sp.dst, sp.R, sp.C, sp.ld, sp.ldp = ptr, buf.data_ptr(), rows, cols, ld, ldp
                    if any(lo <= ptr < hi for lo, hi in _X3_STATIC):
                        if key not in _X3_STATIC_SPLITS:
                            _X3_STATIC_SPLITS[key] = sp
                            new_static.append(sp)
                    elif ptr in presplit and _X3_FRESH.get(key) == _X3_EPOCH[0]:
                        pass                          # an earlier launch of this step left the image (static launch order)
                    else:
                        dyn.append(sp)
                        self._keys.append(key)
                setattr(q, side, buf.data_ptr())
                setattr(q, "lda" if side == "A" else "ldb", 2 * ldp)
            q.flags |= F_KPAD
            out.append(q)
        self.gemm = array(GemmProblem, out)
        self.dyn = array(_lib.SplitProblem, dyn) if dyn else None
        if new_static:                                # first sight of a weight shadow: split it now (plans are built by eager
            _split(_lib.lib(), new_static, _stream())  # launches, never inside a graph capture); later: x3_refresh_static()

    @staticmethod
    def _eligible(p) -> bool:
        al = (p.bias_n or 0) | (p.resid or 0) | (p.C or 0) | (p.gate or 0)
        tm, tn = (p.M + 255) // 256, (p.N + 255) // 256
        return (p.M >= 256 and p.N >= 256 and p.K >= 256 and p.N % 4 == 0 and al % 16 == 0 and (p.ldr | p.ldc | p.ldg) % 4 == 0
                and not p.bias_m and p.splitk <= 1 and not (p.flags & (F_ATOMIC | _lib.F_A_OVERLAP | _lib.F_B_OVERLAP))
                and not (p.resid and (p.flags & F_ACCUM)) and p.M * p.N >= 0.8 * (tm * 256) * (tn * 256)
                and p.lda % 4 == 0 and p.ldb % 4 == 0 and (p.A or 0) % 16 == 0 and (p.B or 0) % 16 == 0)

    def run(self, L, variant: int, seed: int, s: int) -> None:
        if self.dyn is not None:
            for sub, k in _chunks(self.dyn, _lib.SplitProblem, None):
                _lib.check(L.bpm_split_rows(sub, k, s), "bpm_split_rows")
            for key in self._keys:
                _X3_FRESH[key] = _X3_EPOCH[0]
        _lib.check(L.bpm_gemm_grouped(_lib.BPM_BF16X3, variant, self.gemm, len(self.gemm), seed, s), "bpm_gemm_grouped(bf16x3)")


# ----------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------
def attn_problem(Q, K, V, O, ldo, lse, B, H, T, S, dh, dhp, mask_off, *, dO=None, delta=None, dQ=None, lddq=0,
                 dK=None, lddk=0, dV=None, lddv=0, dq_scale=1.0, drop_p=0.0, drop_site=0, q_pos0=0, q_stride=1,
                 dS=None, Pd=None, xs=(0, 0, 0)) -> AttnProblem:
    """dS / Pd (attn_bwd_dq only): CT tensors that receive the score gradient and the dropped probabilities, element
    (b, h, i, j) at b*xs[0] + h*xs[1] + i*xs[2] + j."""
    p = AttnProblem()
    p.q_pos0, p.q_stride = q_pos0, q_stride
    p.dS, p.Pd = _p(dS), _p(Pd)
    p.xs_b, p.xs_h, p.xs_q = xs
    p.Q, p.K, p.V, p.O, p.ldo, p.lse = _p(Q), _p(K), _p(V), _p(O), ldo, _f32(lse, "lse")
    p.dO, p.delta = _p(dO), _f32(delta, "delta")
    p.dQ, p.lddq, p.dK, p.lddk, p.dV, p.lddv = _p(dQ), lddq, _p(dK), lddk, _p(dV), lddv
    p.B, p.H, p.T, p.S, p.dh, p.dhp, p.mask_off = B, H, T, S, dh, dhp, mask_off
    p.dq_scale, p.drop_p, p.drop_site = dq_scale, drop_p, drop_site
    return p


def attn_fwd(dtype: int, probs, seed: int = 0) -> None:
    arr = _as_array(AttnProblem, probs)
    L, s = _lib.lib(), _stream()
    for sub, k in _chunks(arr, AttnProblem, None):
        _lib.check(L.bpm_attn_fwd(dtype, sub, k, _seed(seed), s), "bpm_attn_fwd")


def attn_bwd(dtype: int, probs, seed: int = 0) -> None:
    arr = _as_array(AttnProblem, probs)
    L, s = _lib.lib(), _stream()
    for sub, k in _chunks(arr, AttnProblem, None):
        _lib.check(L.bpm_attn_bwd(dtype, sub, k, _seed(seed), s), "bpm_attn_bwd")


def attn_bwd_dq(dtype: int, probs, seed: int = 0) -> None:
    arr = _as_array(AttnProblem, probs)
    L, s = _lib.lib(), _stream()
    for sub, k in _chunks(arr, AttnProblem, None):
        _lib.check(L.bpm_attn_bwd_dq(dtype, sub, k, _seed(seed), s), "bpm_attn_bwd_dq")


def attn_bwd_dkv(dtype: int, probs, seed: int = 0) -> None:
    arr = _as_array(AttnProblem, probs)
    L, s = _lib.lib(), _stream()
    for sub, k in _chunks(arr, AttnProblem, None):
        _lib.check(L.bpm_attn_bwd_dkv(dtype, sub, k, _seed(seed), s), "bpm_attn_bwd_dkv")


# ----------------------------------------------------------------------------
# row kernels
# ----------------------------------------------------------------------------
def pack_problem(B, T, Cn, ld, *, src=None, dst=None, g=None, ldg=0, dsrc=None, drop_p=0.0, drop_site=0) -> PackProblem:
    p = PackProblem()
    p.src, p.dst, p.g, p.ldg, p.dsrc = _f32(src, "pack.src"), _p(dst), _f32(g, "pack.g"), ldg, _f32(dsrc, "pack.dsrc")
    p.B, p.T, p.C, p.ld, p.drop_p, p.drop_site = B, T, Cn, ld, drop_p, drop_site
    return p


def pack_rows_fwd(dtype, probs, seed=0) -> None:
    arr = _as_array(PackProblem, probs)
    for sub, k in _chunks(arr, PackProblem, None):
        _lib.check(_lib.lib().bpm_pack_rows_fwd(dtype, sub, k, _seed(seed), _stream()), "bpm_pack_rows_fwd")


def pack_rows_bwd(probs, seed=0) -> None:
    arr = _as_array(PackProblem, probs)
    for sub, k in _chunks(arr, PackProblem, None):
        _lib.check(_lib.lib().bpm_pack_rows_bwd(sub, k, _seed(seed), _stream()), "bpm_pack_rows_bwd")


def pack_weights(dtype, table_dev: torch.Tensor, ndesc: int, total_blocks: int) -> None:
    _lib.check(_lib.lib().bpm_pack_weights(dtype, table_dev.data_ptr(), ndesc, total_blocks, _stream()), "bpm_pack_weights")


def device_table(descs) -> torch.Tensor:
    """ctypes descriptor structs -> one device-resident byte tensor (the table-driven launches read it on the GPU)."""
    arr = (type(descs[0]) * len(descs))(*descs)
    t = torch.frombuffer(bytearray(bytes(memoryview(arr))), dtype=torch.uint8)
    return t if _DRY_RUN else t.to("cuda")


def adam_blocks(n4: int) -> int:
    return int(_lib.lib().bpm_adam_blocks(n4)) if not _DRY_RUN else (n4 + 1023) // 1024


def adam_step_table(dtype, table_dev, nseg, nblk, master, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step,
                    grad_scale, zero_grad) -> None:
    for t in (master, grad, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != master.numel():
            raise ValueError("adam_step_table: flat contiguous float32 buffers of one size")
    _lib.check(_lib.lib().bpm_adam_step_table(dtype, table_dev.data_ptr(), nseg, nblk, _p(master), _p(grad), _p(exp_avg), _p(exp_avg_sq),
                                              lr, beta1, beta2, eps, weight_decay, step, grad_scale, int(bool(zero_grad)), _stream()),
               "bpm_adam_step_table")


def fold_bias(table_dev: torch.Tensor, ndesc: int, total_blocks: int) -> None:
    _lib.check(_lib.lib().bpm_fold_bias(table_dev.data_ptr(), ndesc, total_blocks, _stream()), "bpm_fold_bias")


def unfold_grads(table_dev: torch.Tensor, ndesc: int, total_blocks: int, store_dw: bool = False) -> None:
    _lib.check(_lib.lib().bpm_unfold_grads(table_dev.data_ptr(), ndesc, total_blocks, int(store_dw), _stream()), "bpm_unfold_grads")


def zero_table(segments):
    """[(device address, element count)] -> (device table, ndesc, total blocks) for zero_segments."""
    L = _lib.lib()
    descs, blk = [], 0
    for ptr, n in segments:
        z = _lib.ZeroDesc()
        z.p, z.n, z.blk0 = ptr, n, blk
        blk += L.bpm_zero_segment_blocks(n)
        descs.append(z)
    return device_table(descs), len(descs), blk


def zero_segments(table_dev: torch.Tensor, ndesc: int, total_blocks: int) -> None:
    _lib.check(_lib.lib().bpm_zero_segments(table_dev.data_ptr(), ndesc, total_blocks, _stream()), "bpm_zero_segments")


def embed_problem(x, out, T, B, *, accumulate=False, drop_p=0.0, drop_site=0, pos0=0, pos_stride=1) -> EmbedProblem:
    p = EmbedProblem()
    p.x, p.out, p.T, p.B = _f32(x, "embed.x"), _f32(out, "embed.out"), T, B
    p.accumulate, p.drop_p, p.drop_site = int(accumulate), drop_p, drop_site
    p.pos0, p.pos_stride = pos0, pos_stride
    return p


def embed_pos_fwd(probs, table, d, scale, seed=0) -> None:
    arr = _as_array(EmbedProblem, probs)
    if table.dtype != torch.float32 or table.shape[1] != d or not table.is_contiguous():
        raise ValueError("embed_pos_fwd: table must be contiguous fp32 [rows, d]")
    for sub, k in _chunks(arr, EmbedProblem, None):
        _lib.check(_lib.lib().bpm_embed_pos_fwd(sub, k, table.data_ptr(), table.shape[0], d, scale, _seed(seed), _stream()),
                   "bpm_embed_pos_fwd")


def embed_pos_bwd(probs, d, scale, seed=0) -> None:
    arr = _as_array(EmbedProblem, probs)
    for sub, k in _chunks(arr, EmbedProblem, None):
        _lib.check(_lib.lib().bpm_embed_pos_bwd(sub, k, d, scale, _seed(seed), _stream()), "bpm_embed_pos_bwd")


def ln_problem(x, gamma, beta, mean, rstd, R, *, out=None, ldo=0, out_f32=False, dy=None, ldy=0, add=None, dx=None,
               dgamma=None, dbeta=None, cast=None, ldc=0, cast_colsum=None, drop_p=0.0, drop_site=0) -> LnProblem:
    p = LnProblem()
    p.x, p.gamma, p.beta = _f32(x, "ln.x"), _f32(gamma, "ln.gamma"), _f32(beta, "ln.beta")
    p.out, p.ldo, p.out_f32 = _p(out), ldo, int(out_f32)
    p.mean, p.rstd, p.R = _f32(mean, "ln.mean"), _f32(rstd, "ln.rstd"), R
    p.dy, p.ldy, p.add, p.dx = _f32(dy, "ln.dy"), ldy, _f32(add, "ln.add"), _f32(dx, "ln.dx")
    p.dgamma, p.dbeta = _f32(dgamma, "ln.dgamma"), _f32(dbeta, "ln.dbeta")
    p.cast, p.ldc = _p(cast), ldc
    p.cast_colsum = cast_colsum if isinstance(cast_colsum, int) else _f32(cast_colsum, "ln.cast_colsum")
    p.drop_p, p.drop_site = drop_p, drop_site
    return p


def ln_fwd(dtype, probs, d, eps=1e-5) -> None:
    arr = _as_array(LnProblem, probs)
    for sub, k in _chunks(arr, LnProblem, None):
        _lib.check(_lib.lib().bpm_ln_fwd(dtype, sub, k, d, eps, _stream()), "bpm_ln_fwd")


def check_ln_rows(probs, d: int) -> None:
    """Host-side contract of bpm_ln_bwd_ws for one launch group, checked where the launch TABLES are built (not per
    step): the dgamma / dbeta / cast_colsum rows ([d] floats each) of its problems are pairwise disjoint.  The last block
    of a problem adds its partial sums with a plain read-modify-write; the library detects EQUAL row pointers and falls
    back to atomics, but rows that overlap partially would be summed wrongly without any error."""
    rows = []
    for i, p in enumerate(probs):
        for nm in ("dgamma", "dbeta", "cast_colsum"):
            a = getattr(p, nm)
            if a and (nm != "cast_colsum" or p.cast):
                rows.append((int(a), i, nm))
    rows.sort()
    for (a, i, n1), (b, j, n2) in zip(rows, rows[1:]):
        if b < a + 4 * d:
            raise ValueError(f"ln_bwd group: {n1} of problem {i} and {n2} of problem {j} share gradient words "
                             f"(0x{a:x}, 0x{b:x}; rows are {4 * d} bytes): one launch must own each row exactly once")


_LN_WS = {}


def _ln_workspace(n: int, d: int, device) -> torch.Tensor:
    """Per-(device, stream) partial-sum workspace of bpm_ln_bwd_ws: launches on one stream are serialised, so they
    can share it; the side stream gets its own."""
    key = (str(device), _stream())
    need = _lib.lib().bpm_ln_bwd_ws_bytes(n, d)
    t = _LN_WS.get(key)
    if t is None or t.numel() * 4 < need:
        t = torch.zeros((need + 3) // 4, device=device, dtype=torch.float32)      # ticket words must start at zero
        _LN_WS[key] = t
    return t


def ln_bwd(probs, d, dtype=None, seed=0) -> None:
    """dtype / seed only matter for problems with a fused `cast` output.  Parameter / bias gradients go through a
    partial-sum workspace (no float atomics)."""
    arr = _as_array(LnProblem, probs)
    dt = BPM_F32 if dtype is None else dtype
    dev = torch.device("cuda", torch.cuda.current_device())
    for sub, k in _chunks(arr, LnProblem, None):
        ws = _ln_workspace(k, d, dev)
        try:
            _lib.check(_lib.lib().bpm_ln_bwd_ws(dt, sub, k, d, _seed(seed), ws.data_ptr(), ws.numel() * 4, _stream()), "bpm_ln_bwd_ws")
        except RuntimeError:
            _LN_WS.clear()          # an aborted launch may leave ticket words set: start from fresh zeroed workspaces
            raise


def cast_problem(a, lda, R, Cn, *, a_is_ct=False, b=None, ldb=0, dst_ct=None, ldd=0, ct_cols=0, dst_f32=None, ldf=0, colsum=None,
                 drop_p=0.0, drop_site=0) -> CastProblem:
    p = CastProblem()
    p.a, p.lda, p.a_is_ct = _p(a), lda, int(a_is_ct)
    p.b, p.ldb = _f32(b, "cast.b"), ldb
    p.dst_ct, p.ldd, p.ct_cols, p.dst_f32, p.ldf = _p(dst_ct), ldd, ct_cols, _f32(dst_f32, "cast.dst_f32"), ldf
    p.colsum = colsum if isinstance(colsum, int) else _f32(colsum, "cast.colsum")
    p.R, p.C, p.drop_p, p.drop_site = R, Cn, drop_p, drop_site
    return p


def rows_cast(dtype, probs, seed=0) -> None:
    arr = _as_array(CastProblem, probs)
    for sub, k in _chunks(arr, CastProblem, None):
        _lib.check(_lib.lib().bpm_rows_cast(dtype, sub, k, _seed(seed), _stream()), "bpm_rows_cast")


def addn_problem(out, ins) -> "_lib.AddnProblem":
    """out = sum(ins): contiguous fp32 tensors of one size (16-byte aligned); out may be one of them."""
    p = _lib.AddnProblem()
    n = out.numel()
    if not 1 <= len(ins) <= 8:
        raise ValueError("add_n takes 1 to 8 inputs")
    for t in (out, *ins):
        if t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
            raise ValueError("add_n: contiguous float32 tensors of one size")
    p.out, p.n_in, p.count = _p(out), len(ins), n
    for j, t in enumerate(ins):
        p.src[j] = _p(t)
    return p


def expand_problem(q, dO, qexp, dOexp, B, H, T, dh, dhp, ld, *, Pd=None, S=0, dbias=None) -> "_lib.ExpandProblem":
    """bpm_expand_heads: head-major q / dO [B,H,T,dhp] -> block rows [(h*T+t)*B+b, ld]; dbias[H*dh] (written) =
    sum_{b,t} rowsum(Pd row) * dO[b,h,t,:]."""
    p = _lib.ExpandProblem()
    p.q, p.dO, p.qexp, p.dOexp, p.Pd, p.dbias = _p(q), _p(dO), _p(qexp), _p(dOexp), _p(Pd), _f32(dbias, "expand.dbias")
    p.B, p.H, p.T, p.S, p.dh, p.dhp, p.ld = B, H, T, S, dh, dhp, ld
    return p


def expand_heads(dtype: int, probs) -> None:
    arr = _as_array(_lib.ExpandProblem, probs)
    for sub, k in _chunks(arr, _lib.ExpandProblem, None):
        _lib.check(_lib.lib().bpm_expand_heads(dtype, sub, k, _stream()), "bpm_expand_heads")


def add_n(probs) -> None:
    arr = _as_array(_lib.AddnProblem, probs)
    for sub, k in _chunks(arr, _lib.AddnProblem, None):
        _lib.check(_lib.lib().bpm_add_n(sub, k, _stream()), "bpm_add_n")


def gmu_problem(a1, a2, ag, x1, x2, R, *, out=None, dout=None, da1=None, da2=None, dag=None, ldg=0, dx1=None, dx2=None) -> GmuProblem:
    p = GmuProblem()
    p.a1, p.a2, p.ag, p.x1, p.x2 = (_f32(t, "gmu") for t in (a1, a2, ag, x1, x2))
    p.out, p.dout = _f32(out, "gmu.out"), _f32(dout, "gmu.dout")
    p.da1, p.da2, p.dag, p.ldg = _p(da1), _p(da2), _p(dag), ldg
    p.dx1, p.dx2, p.R = _f32(dx1, "gmu.dx1"), _f32(dx2, "gmu.dx2"), R
    return p


def gmu2_fwd(probs, d) -> None:
    arr = _as_array(GmuProblem, probs)
    for sub, k in _chunks(arr, GmuProblem, None):
        _lib.check(_lib.lib().bpm_gmu2_fwd(sub, k, d, _stream()), "bpm_gmu2_fwd")


def gmu2_bwd(dtype, probs, d) -> None:
    arr = _as_array(GmuProblem, probs)
    for sub, k in _chunks(arr, GmuProblem, None):
        _lib.check(_lib.lib().bpm_gmu2_bwd(dtype, sub, k, d, _stream()), "bpm_gmu2_bwd")


# ----------------------------------------------------------------------------
# [B,d] tail: token pick + n-way gated fusion + residual head
# ----------------------------------------------------------------------------
def tail_fwd(desc, seed: int = 0) -> None:
    _lib.check(_lib.lib().bpm_tail_fwd(C.byref(desc), _seed(seed), _stream()), "bpm_tail_fwd")


def tail_bwd(desc, grads) -> None:
    _lib.check(_lib.lib().bpm_tail_bwd(C.byref(desc), C.byref(grads), _stream()), "bpm_tail_bwd")
