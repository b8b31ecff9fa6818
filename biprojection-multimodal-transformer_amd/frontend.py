"""Front-ends of the 4-modal model on the HIP path: the AudioEncoder convolution stack
(mmtr.py:93-108) and the poster projection (mmtr.py:310, 486).  Both sit immediately upstream of the trunk; their
parameters stay ordinary nn.Parameters (reference names `audio_enc.conv_layers.{0,1}.{weight,bias}`,
`proj_poster.weight`) and their gradients are returned through autograd.

AudioEncoder = Conv1d(96, 96, k=128, stride=2) x 2 + AdaptiveAvgPool1d(out).  Each convolution is an IMPLICIT product on
the grouped MFMA GEMM: the signal is kept channels-last, xc[(b, pos), ci], and window l of a batch element is the
contiguous run of taps*Cin elements starting at row stride*l -- a matrix whose rows overlap (leading dimension stride*Cin,
row length K = taps*Cin = 12288; BPM_GEMM_A_OVERLAP / _B_OVERLAP of include/bpmult_hip.h).  No window matrix is written:

  forward          y_v[r, co]  = xc-windows . Wr^T + bias          NT   (Wr[co, k*Cin + ci] = W[co, ci, k])
  weight gradient  dWr         = dy_v^T . xc-windows (+ bias sums) TN   (the same xc buffer as the overlapping B operand)
  data gradient    dx[(b, s*m + p), ci] = dyp-windows . Wt_p^T     NT   one problem per phase p = pos % stride over the
                   zero-padded output gradient, Wt_p[ci, jj*Cout + co] = W[co, ci, s*(J-1-jj) + p], J = taps / stride

The rows of all batch elements share one stride (a batch element's positions are trimmed to taps + stride*(Lout-1), a
multiple of the stride), so every product is ONE problem over "virtual" rows: the taps/stride - 1 rows per batch element
whose windows straddle the next element are computed and dropped (3 % of the rows at T = 4096; their output-gradient
rows are zeros).  HIP kernels for the layout changes (bpm_signal_pack / _unpack) and the adaptive pooling; activations
are [(b, position), channel] matrices throughout, and the pooled [B*out, 96] result IS the [B, out, 96] tensor the model
feeds to `proj_a` (mmtr.py:449 transposes to it).
"""
from __future__ import annotations

import torch

from . import _lib, config, ops
from ._lib import F_A_OVERLAP, F_ACCUM, F_B_OVERLAP, F_KPAD, GEMM_NN, GEMM_NT, GEMM_TN, OUT_F32
from .ops import pad32


def _s():
    return torch.cuda.current_stream().cuda_stream


def _ct_copy(w2d: torch.Tensor, dtype: int) -> torch.Tensor:
    """CT copy of a [rows, cols] fp32 weight (cols % 32 == 0 here: 12288 / 6144 / 4096)."""
    rows, cols = w2d.shape
    w2d = w2d.contiguous()
    out = torch.empty(rows, pad32(cols), device=w2d.device, dtype=ops.ct_torch(dtype))
    ops.rows_cast(dtype, [ops.cast_problem(w2d, cols, rows, cols, dst_ct=out, ldd=pad32(cols))], 0)
    return out


def _pack(dtype, x, B, C, L, strides, front, rows_per_batch, total_rows):
    """CT channels-last rows of the fp32 signal x[b*sb + c*sc + l*sl]: [total_rows, C], zeros outside the L valid rows."""
    out = torch.empty(total_rows, C, device=x.device, dtype=ops.ct_torch(dtype))
    sb, sc, sl = strides
    _lib.check(_lib.lib().bpm_signal_pack(dtype, x.data_ptr(), out.data_ptr(), B, C, L, sb, sc, sl, front, rows_per_batch, total_rows, C, _s()),
               "bpm_signal_pack")
    return out


def _unpack(src, dst, B, C, L, strides, Lvalid, rows_per_batch, ld):
    sb, sc, sl = strides
    _lib.check(_lib.lib().bpm_signal_unpack(src.data_ptr(), dst.data_ptr(), B, C, L, sb, sc, sl, Lvalid, rows_per_batch, ld, _s()), "bpm_signal_unpack")
    return dst


class _Conv1dFn(torch.autograd.Function):
    """y[(b,l), co] = sum_{ci,k} W[co,ci,k] x[b,ci,stride*l+k] + bias[co];  x given with element strides (sb, sc, sl)."""

    @staticmethod
    def forward(ctx, x, weight, bias, B, Lin, strides, stride, dtype):
        Cout, Cin, K = weight.shape
        Lout = (Lin - K) // stride + 1
        if Lout < 1:
            raise ValueError(f"AudioEncoder: sequence of {Lin} frames is shorter than the {K}-tap kernel")
        sz = 2 if dtype == _lib.BPM_BF16 else 4
        J = K // stride
        if (K % stride or Cin % 4 or Cout % 4 or (stride * Cin * sz) % 16 or (Cout * sz) % 16 or (K * Cin) % 64 or (J * Cout) % 64):
            raise ValueError(f"Conv1d({Cin},{Cout},k={K},stride={stride}): the implicit product needs taps % stride == 0 and 64-element "
                             f"window rows (the reference's AudioEncoder is Conv1d(96,96,128,stride=2))")
        Lp = K + stride * (Lout - 1)              # positions any window reads: a multiple of the stride
        Lv = Lp // stride                         # virtual rows per batch element (Lout valid + J - 1 straddling)
        Mv = B * Lv
        Mv_pad = (Mv + 63) // 64 * 64             # k extent of the weight-gradient product
        rows = max(B * Lp + K - stride, (Mv_pad - 1) * stride + K)      # readable through the last (virtual / padded) window
        xc = _pack(dtype, x, B, Cin, Lp, strides, 0, Lp, rows)
        wr = _ct_copy(weight.detach().permute(0, 2, 1).reshape(Cout, K * Cin), dtype)          # k major
        yv = torch.empty(Mv, Cout, device=x.device, dtype=torch.float32)
        ops.gemm_grouped(dtype, GEMM_NT, [ops.gemm_problem(xc, wr, yv, Mv, Cout, K * Cin, stride * Cin, wr.shape[1], Cout, bias_n=bias.detach(),
                                                           out_kind=OUT_F32, flags=F_KPAD | F_A_OVERLAP)], 0)
        y = _unpack(yv, torch.empty(B * Lout, Cout, device=x.device), B, Cout, Lout, (Lout * Cout, 1, Cout), Lout, Lv, Cout)
        ctx.save_for_backward(xc, weight.detach())
        ctx.meta = (B, Cin, K, stride, Lin, Lout, Cout, strides, dtype, x.shape, x.requires_grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, weight = ctx.saved_tensors
        B, Cin, K, stride, Lin, Lout, Cout, strides, dtype, xshape, need_dx = ctx.meta
        J = K // stride
        Lp = K + stride * (Lout - 1)
        Lv = Lp // stride
        Mv_pad = (B * Lv + 63) // 64 * 64
        dy = dy.contiguous().float()
        dy_strides = (Lout * Cout, 1, Cout)
        # weight gradient: dWr[co, (k, ci)] = sum over virtual rows of dy_v[r, co] * window_r[(k, ci)]; straddling rows carry zeros
        dyv = _pack(dtype, dy, B, Cout, Lout, dy_strides, 0, Lv, Mv_pad)
        dWr = torch.empty(Cout, K * Cin, device=dy.device)
        db = torch.zeros(Cout, device=dy.device)
        ops.gemm_grouped(dtype, GEMM_TN, [ops.gemm_problem(dyv, xc, dWr, Cout, K * Cin, Mv_pad, Cout, stride * Cin, K * Cin,
                                                           flags=F_KPAD | F_B_OVERLAP, colsum_a=db)], 0)
        dW = dWr.view(Cout, K, Cin).permute(0, 2, 1).contiguous()
        dx = None
        if need_dx:
            # dx[(b, s*m + p), ci] = sum_{jj, co} dyp[(b, m + jj), co] * W[co, ci, s*(J-1-jj) + p]: dyp = dy behind J-1 zero rows
            Pb = Lv + J - 1
            dyp = _pack(dtype, dy, B, Cout, Lout, dy_strides, J - 1, Pb, B * Pb + J - 1)
            w4 = weight.view(Cout, Cin, J, stride).flip(2)                                  # [co, ci, jj, p]
            dxv = torch.empty(B * Pb, stride * Cin, device=dy.device)
            probs, keep = [], []
            for p in range(stride):
                wt = _ct_copy(w4[..., p].permute(1, 2, 0).reshape(Cin, J * Cout), dtype)
                keep.append(wt)
                probs.append(ops.gemm_problem(dyp, wt, dxv[:, p * Cin:], B * Pb, Cin, J * Cout, Cout, wt.shape[1], stride * Cin,
                                              out_kind=OUT_F32, flags=F_KPAD | F_A_OVERLAP))
            ops.gemm_grouped(dtype, GEMM_NT, probs, 0)
            dx = _unpack(dxv, torch.empty(xshape, device=dy.device), B, Cin, Lin, strides, Lp, Pb * stride, Cin)
        return dx, dW, db, None, None, None, None, None


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
