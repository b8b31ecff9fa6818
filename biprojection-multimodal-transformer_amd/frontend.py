"""Front-ends of the 4-modal model on the HIP path: the AudioEncoder convolution stack
(mmtr.py:93-108) and the poster projection (mmtr.py:310, 486).  Both sit immediately upstream of the trunk; their
parameters stay ordinary nn.Parameters (reference names `audio_enc.conv_layers.{0,1}.{weight,bias}`,
`proj_poster.weight`) and their gradients are returned through autograd.

AudioEncoder = Conv1d(96, 96, k=128, stride=2) x 2 + AdaptiveAvgPool1d(out): each convolution is a product over window
rows (K = 96 * 128 = 12288) on the grouped MFMA GEMM -- forward NT with the bias, weight gradient TN with the bias
column sums, data gradient NN -- with HIP kernels for the window gather (bpm_im2col1d), its adjoint (bpm_col2im1d,
a gather: no atomics) and the adaptive pooling; activations are [(b, position), channel] matrices throughout, and the
pooled [B*out, 96] result IS the [B, out, 96] tensor the model feeds to `proj_a` (mmtr.py:449 transposes to it).
"""
from __future__ import annotations

import torch

from . import _lib, config, ops
from ._lib import F_ACCUM, F_KPAD, GEMM_NN, GEMM_NT, GEMM_TN, OUT_F32
from .ops import pad32


def _s():
    return torch.cuda.current_stream().cuda_stream


def _ct_copy(w2d: torch.Tensor, dtype: int) -> torch.Tensor:
    """CT copy of a [rows, cols] fp32 weight (cols % 32 == 0 here: 12288 / 4096)."""
    rows, cols = w2d.shape
    out = torch.empty(rows, pad32(cols), device=w2d.device, dtype=ops.ct_torch(dtype))
    ops.rows_cast(dtype, [ops.cast_problem(w2d, cols, rows, cols, dst_ct=out, ldd=pad32(cols))], 0)
    return out


class _Conv1dFn(torch.autograd.Function):
    """y[(b,l), co] = sum_{ci,k} W[co,ci,k] x[b,ci,stride*l+k] + bias[co];  x given with element strides (sb, sc, sl)."""

    @staticmethod
    def forward(ctx, x, weight, bias, B, Lin, strides, stride, dtype):
        L = _lib.lib()
        Cout, Cin, K = weight.shape
        Lout = (Lin - K) // stride + 1
        if Lout < 1:
            raise ValueError(f"AudioEncoder: sequence of {Lin} frames is shorter than the {K}-tap kernel")
        ck = Cin * K
        ldc = pad32(ck)
        ct = ops.ct_torch(dtype)
        col = torch.zeros(B * Lout, ldc, device=x.device, dtype=ct) if ldc != ck else torch.empty(B * Lout, ldc, device=x.device, dtype=ct)
        sb, sc, sl = strides
        _lib.check(L.bpm_im2col1d(dtype, x.data_ptr(), col.data_ptr(), B, Cin, K, stride, Lin, Lout, sb, sc, sl, ldc, _s()), "bpm_im2col1d")
        wct = _ct_copy(weight.detach().reshape(Cout, ck), dtype)
        y = torch.empty(B * Lout, Cout, device=x.device, dtype=torch.float32)
        ops.gemm_grouped(dtype, GEMM_NT, [ops.gemm_problem(col, wct, y, B * Lout, Cout, ck, ldc, ldc, Cout, bias_n=bias.detach(),
                                                           out_kind=OUT_F32, flags=F_KPAD)], 0)
        ctx.save_for_backward(col, wct)
        ctx.meta = (B, Cin, K, stride, Lin, Lout, Cout, strides, dtype, x.shape, x.requires_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        col, wct = ctx.saved_tensors
        B, Cin, K, stride, Lin, Lout, Cout, strides, dtype, xshape, need_dx = ctx.meta
        L = _lib.lib()
        ck, ldc, R = Cin * K, col.shape[1], B * Lout
        dy = dy.contiguous().float()
        ldd = pad32(Cout)
        dyc = torch.zeros(R, ldd, device=dy.device, dtype=col.dtype)
        ops.rows_cast(dtype, [ops.cast_problem(dy, Cout, R, Cout, dst_ct=dyc, ldd=ldd)], 0)
        dW = torch.zeros(Cout, ck, device=dy.device)
        db = torch.zeros(Cout, device=dy.device)
        ops.gemm_grouped(dtype, GEMM_TN, [ops.gemm_problem(dyc, col, dW, Cout, ck, R, ldd, ldc, ck, flags=F_ACCUM, colsum_a=db)], 0)
        dx = None
        if need_dx:
            dcol = torch.empty(R, ldc, device=dy.device)
            ops.gemm_grouped(dtype, GEMM_NN, [ops.gemm_problem(dyc, wct, dcol, R, ck, Cout, ldd, ldc, ldc, flags=F_KPAD)], 0)
            dx = torch.empty(xshape, device=dy.device)
            sb, sc, sl = strides
            _lib.check(L.bpm_col2im1d(dcol.data_ptr(), dx.data_ptr(), B, Cin, K, stride, Lin, Lout, sb, sc, sl, ldc, 0, _s()), "bpm_col2im1d")
        return dx, dW.view(Cout, Cin, K), db, None, None, None, None, None


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, B, C, Lin, Lout):
        out = torch.empty(B * Lout, C, device=y.device)
        _lib.check(_lib.lib().bpm_adaptive_pool1d_fwd(y.data_ptr(), out.data_ptr(), B, C, Lin, Lout, _s()), "bpm_adaptive_pool1d_fwd")
        ctx.meta = (B, C, Lin, Lout)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C, Lin, Lout = ctx.meta
        dout = dout.contiguous().float()
        dy = torch.empty(B * Lin, C, device=dout.device)
        _lib.check(_lib.lib().bpm_adaptive_pool1d_bwd(dout.data_ptr(), dy.data_ptr(), B, C, Lin, Lout, _s()), "bpm_adaptive_pool1d_bwd")
        return dy, None, None, None, None


def audio_encoder_forward(audio: torch.Tensor, convs, pool_out: int, precision: str) -> torch.Tensor:
    """audio fp32 [B, 96, T_a] -> [B, pool_out, 96] (= AudioEncoder(audio).transpose(1, 2) of mmtr.py:449)."""
    if not audio.is_cuda:
        raise RuntimeError("AudioEncoder: the HIP front-end needs CUDA (HIP) tensors; there is no CPU path")
    dtype = config.dtype_code(precision)
    B, C, Ta = audio.shape
    x = audio.contiguous().float()
    c1, c2 = convs
    y1 = _Conv1dFn.apply(x, c1.weight, c1.bias, B, Ta, (C * Ta, Ta, 1), c1.stride[0], dtype)            # [(b,l1), 96]
    L1 = y1.shape[0] // B
    C1 = y1.shape[1]
    y2 = _Conv1dFn.apply(y1, c2.weight, c2.bias, B, L1, (L1 * C1, 1, C1), c2.stride[0], dtype)          # reads [(b,l), c] in place
    L2 = y2.shape[0] // B
    out = _PoolFn.apply(y2, B, y2.shape[1], L2, pool_out)
    return out.view(B, pool_out, y2.shape[1])


class _SkinnyLinearFn(torch.autograd.Function):
    """out[B, N] = x[B, K] W[N, K]^T (no bias) on the grouped GEMM: the poster projection (mmtr.py:310, 486)."""

    @staticmethod
    def forward(ctx, x, weight, dtype):
        Bn, K = x.shape
        N = weight.shape[0]
        ldk = pad32(K)
        xc = torch.zeros(Bn, ldk, device=x.device, dtype=ops.ct_torch(dtype))
        ops.rows_cast(dtype, [ops.cast_problem(x.contiguous().float(), K, Bn, K, dst_ct=xc, ldd=ldk)], 0)
        wct = _ct_copy(weight.detach(), dtype)
        out = torch.empty(Bn, N, device=x.device)
        ops.gemm_grouped(dtype, GEMM_NT, [ops.gemm_problem(xc, wct, out, Bn, N, K, ldk, wct.shape[1], N, flags=F_KPAD)], 0)
        ctx.save_for_backward(xc, wct)
        ctx.meta = (Bn, K, N, dtype, x.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dout):
        xc, wct = ctx.saved_tensors
        Bn, K, N, dtype, need_dx = ctx.meta
        dout = dout.contiguous().float()
        ldn = pad32(N)
        dc = torch.zeros(Bn, ldn, device=dout.device, dtype=xc.dtype)
        ops.rows_cast(dtype, [ops.cast_problem(dout, N, Bn, N, dst_ct=dc, ldd=ldn)], 0)
        dW = torch.zeros(N, K, device=dout.device)
        ops.gemm_grouped(dtype, GEMM_TN, [ops.gemm_problem(dc, xc, dW, N, K, Bn, ldn, xc.shape[1], K, flags=F_ACCUM)], 0)
        dx = None
        if need_dx:
            dx = torch.empty(Bn, K, device=dout.device)
            ops.gemm_grouped(dtype, GEMM_NN, [ops.gemm_problem(dc, wct, dx, Bn, K, N, ldn, wct.shape[1], K, flags=F_KPAD)], 0)
        return dx, dW, None


def skinny_linear(x: torch.Tensor, weight: torch.Tensor, precision: str) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("poster projection: the HIP front-end needs CUDA (HIP) tensors; there is no CPU path")
    return _SkinnyLinearFn.apply(x, weight, config.dtype_code(precision))
