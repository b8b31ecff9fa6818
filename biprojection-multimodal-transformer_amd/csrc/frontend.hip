// Front-end of the 4-modal model that feeds the hot path (gfx950): AudioEncoder (mmtr.py:93-108) =
// Conv1d(96, 96, k=128, stride 2) x 2 + AdaptiveAvgPool1d(200), SURVEY.md 8(f) rank 2.
//
// A 1-D convolution with a 128-tap kernel over 96 channels is a GEMM with K = 96 * 128 = 12288:
//   y[(b,l), co] = sum_{ci,k} W[co, ci*128 + k] * x[b, ci, stride*l + k] + bias[co]
// The products run on the grouped MFMA GEMM of this library (gemm.hip: forward NT, weight gradient TN with the bias
// column sums, data gradient NN); this file holds what is specific to the convolution:
//   bpm_im2col1d   : window rows  col[(b,l), ci*K + k] = x[b*sb + ci*sc + (stride*l + k)*sl]  (CT, leading dim padded)
//                    -- arbitrary element strides, so the second layer reads the first layer's [(b,l), c] output
//                    directly and nothing is transposed anywhere
//   bpm_col2im1d   : its adjoint as a GATHER (no atomics): dx[b, ci, p] = sum over the <= K/stride windows that cover p
//   bpm_adaptive_pool1d_{fwd,bwd}: AdaptiveAvgPool1d over the position axis of a [(b,l), c] matrix; the [B*out, C]
//                    result IS the [B, 200, 96] tensor the model transposes to (mmtr.py:449)
// All are streaming kernels (HBM-bound); an explicit window matrix costs 2 x 86 MB of traffic per layer-1 call at
// batch 8 (35 us at HBM rate) against ~8 GFLOP of products.
#include "bpm_common.h"
#include "../../include/bpmult_hip.h"

namespace {

constexpr int FNT = 256;

template <typename CT>
__global__ __launch_bounds__(FNT) void im2col1d_kernel(const float* __restrict__ x, CT* __restrict__ col, int B, int Cin, int K, int stride,
                                                       int Lout, long sb, long sc, long sl, int ldcol) {
    // one thread = 4 consecutive taps k of one (row, ci); K % 4 == 0
    const long idx = (long)blockIdx.x * FNT + threadIdx.x;
    const int kq = K >> 2;
    const long per_row = (long)Cin * kq;
    const long rows = (long)B * Lout;
    if (idx >= rows * per_row) return;
    const long row = idx / per_row;
    const int rem = (int)(idx % per_row), ci = rem / kq, k = (rem % kq) * 4;
    const int b = (int)(row / Lout), l = (int)(row % Lout);
    const float* p = x + b * sb + ci * sc + ((long)stride * l + k) * sl;
    const float v0 = p[0], v1 = p[sl], v2 = p[2 * sl], v3 = p[3 * sl];
    CT* o = col + row * ldcol + ci * K + k;
    if constexpr (sizeof(CT) == 4) *(f32x4*)o = f32x4{v0, v1, v2, v3};
    else { bf16x4 t; t[0] = (bf16_t)v0; t[1] = (bf16_t)v1; t[2] = (bf16_t)v2; t[3] = (bf16_t)v3; *(bf16x4*)o = t; }
}

// dx[b, ci, p] (+)= sum_{l : 0 <= p - stride*l < K, l < Lout} dcol[(b,l), ci*K + p - stride*l]
__global__ __launch_bounds__(FNT) void col2im1d_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int Cin, int K, int stride,
                                                       int Lout, int Lin, long sb, long sc, long sl, int ldcol, int accumulate) {
    const long idx = (long)blockIdx.x * FNT + threadIdx.x;
    if (idx >= (long)B * Cin * Lin) return;
    const int p = (int)(idx % Lin), ci = (int)((idx / Lin) % Cin), b = (int)(idx / ((long)Lin * Cin));
    int lhi = p / stride;
    if (lhi > Lout - 1) lhi = Lout - 1;
    int llo = (p - K + stride) / stride;                 // smallest l with p - stride*l <= K - 1
    if (p - K + 1 <= 0) llo = 0;
    float s = 0.f;
    for (int l = llo; l <= lhi; ++l) {
        const int k = p - stride * l;
        if (k >= 0 && k < K) s += dcol[((long)b * Lout + l) * ldcol + ci * K + k];
    }
    float* o = dx + b * sb + ci * sc + (long)p * sl;
    *o = accumulate ? *o + s : s;
}

BPM_DEV void pool_window(int i, int Lin, int Lout, int& lo, int& hi) {       // torch adaptive pooling: [floor(i L / O), ceil((i+1) L / O))
    lo = (int)(((long)i * Lin) / Lout);
    hi = (int)((((long)(i + 1)) * Lin + Lout - 1) / Lout);
}

// out[(b,i), c] = mean_{l in window(i)} y[(b,l), c]
__global__ __launch_bounds__(FNT) void pool_fwd_kernel(const float* __restrict__ y, float* __restrict__ out, int B, int C, int Lin, int Lout) {
    const long idx = (long)blockIdx.x * FNT + threadIdx.x;
    if (idx >= (long)B * Lout * C) return;
    const int c = (int)(idx % C), i = (int)((idx / C) % Lout), b = (int)(idx / ((long)C * Lout));
    int lo, hi;
    pool_window(i, Lin, Lout, lo, hi);
    float s = 0.f;
    for (int l = lo; l < hi; ++l) s += y[((long)b * Lin + l) * C + c];
    out[idx] = s / (float)(hi - lo);
}

// dy[(b,l), c] = sum_{i : l in window(i)} dout[(b,i), c] / |window(i)|
__global__ __launch_bounds__(FNT) void pool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dy, int B, int C, int Lin, int Lout) {
    const long idx = (long)blockIdx.x * FNT + threadIdx.x;
    if (idx >= (long)B * Lin * C) return;
    const int c = (int)(idx % C), l = (int)((idx / C) % Lin), b = (int)(idx / ((long)C * Lin));
    // windows that contain l: i from floor(l O / L) - 1 (clamped) upwards while lo(i) <= l
    int i0 = (int)(((long)l * Lout) / Lin) - 1;
    if (i0 < 0) i0 = 0;
    float s = 0.f;
    for (int i = i0; i < Lout; ++i) {
        int lo, hi;
        pool_window(i, Lin, Lout, lo, hi);
        if (lo > l) break;
        if (l < hi) s += dout[((long)b * Lout + i) * C + c] / (float)(hi - lo);
    }
    dy[idx] = s;
}

}  // namespace

extern "C" int bpm_im2col1d(int dtype, const float* x, void* col, int B, int Cin, int K, int stride, int Lin, int Lout,
                            int64_t sb, int64_t sc, int64_t sl, int ldcol, void* stream) {
    if (!x || !col || B < 1 || Cin < 1 || K < 4 || (K & 3) || stride < 1 || Lout < 1 || ldcol < Cin * K) return BPM_ERR_ARG;
    if ((long)stride * (Lout - 1) + K > Lin) return BPM_ERR_ARG;
    if (dtype != BPM_F32 && dtype != BPM_BF16) return BPM_ERR_ARG;
    if (((uintptr_t)col & 15) || (ldcol & 3)) return BPM_ERR_ALIGN;
    const long n = (long)B * Lout * Cin * (K >> 2);
    const dim3 grid((unsigned)((n + FNT - 1) / FNT));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == BPM_BF16) hipLaunchKernelGGL(im2col1d_kernel<bf16_t>, grid, dim3(FNT), 0, s, x, (bf16_t*)col, B, Cin, K, stride, Lout, sb, sc, sl, ldcol);
    else hipLaunchKernelGGL(im2col1d_kernel<float>, grid, dim3(FNT), 0, s, x, (float*)col, B, Cin, K, stride, Lout, sb, sc, sl, ldcol);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_col2im1d(const float* dcol, float* dx, int B, int Cin, int K, int stride, int Lin, int Lout,
                            int64_t sb, int64_t sc, int64_t sl, int ldcol, int accumulate, void* stream) {
    if (!dcol || !dx || B < 1 || Cin < 1 || K < 1 || stride < 1 || Lout < 1 || Lin < 1 || ldcol < Cin * K) return BPM_ERR_ARG;
    if ((long)stride * (Lout - 1) + K > Lin) return BPM_ERR_ARG;
    const long n = (long)B * Cin * Lin;
    hipLaunchKernelGGL(col2im1d_kernel, dim3((unsigned)((n + FNT - 1) / FNT)), dim3(FNT), 0, (hipStream_t)stream, dcol, dx, B, Cin, K, stride,
                       Lout, Lin, sb, sc, sl, ldcol, accumulate);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_adaptive_pool1d_fwd(const float* y, float* out, int B, int C, int Lin, int Lout, void* stream) {
    if (!y || !out || B < 1 || C < 1 || Lin < 1 || Lout < 1) return BPM_ERR_ARG;
    const long n = (long)B * Lout * C;
    hipLaunchKernelGGL(pool_fwd_kernel, dim3((unsigned)((n + FNT - 1) / FNT)), dim3(FNT), 0, (hipStream_t)stream, y, out, B, C, Lin, Lout);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_adaptive_pool1d_bwd(const float* dout, float* dy, int B, int C, int Lin, int Lout, void* stream) {
    if (!dout || !dy || B < 1 || C < 1 || Lin < 1 || Lout < 1) return BPM_ERR_ARG;
    const long n = (long)B * Lin * C;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)((n + FNT - 1) / FNT)), dim3(FNT), 0, (hipStream_t)stream, dout, dy, B, C, Lin, Lout);
    BPM_CHECK_LAUNCH();
    return 0;
}
