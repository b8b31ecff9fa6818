// Front-end of the 4-modal model that feeds the hot path (gfx950): AudioEncoder (mmtr.py:93-108) =
// Conv1d(96, 96, k=128, stride 2) x 2 + AdaptiveAvgPool1d(200), SURVEY.md 8(f) rank 2.
//
// A strided 1-D convolution over a CHANNELS-LAST signal xc[(b, pos), ci] is a product whose left operand is the signal
// itself read with overlapping rows: window l of a batch element is the contiguous run of taps*Cin elements that starts at
// row stride*l, so
//   y[(b,l), co] = sum_{k,ci} Wr[co, k*Cin + ci] * xc[(b, stride*l + k), ci] + bias[co]        (Wr = W with k major)
// is bpm_gemm_grouped NT with lda = stride*Cin < K = taps*Cin (BPM_GEMM_A_OVERLAP), the weight gradient is TN with the
// same buffer as the overlapping B operand, and the data gradient is the same form again over the zero-padded output
// gradient (one problem per output phase pos % stride; frontend.py).  No window matrix exists anywhere (the explicit
// one was taps/stride = 64x the signal: 2 x 86 MB per layer-1 call at batch 8).  This file holds what is left:
//   bpm_signal_pack  : fp32 signal with arbitrary element strides -> CT channels-last rows, `front` zero rows ahead of
//                      each batch element's L rows and zeros up to `rows_per_batch` / `total_rows`
//   bpm_signal_unpack: fp32 rows [(b, l), c] (leading dim, rows per batch) -> fp32 signal with arbitrary strides
//   bpm_adaptive_pool1d_{fwd,bwd}: AdaptiveAvgPool1d over the position axis of a [(b,l), c] matrix; the [B*out, C]
//                      result IS the [B, 200, 96] tensor the model transposes to (mmtr.py:449)
// All are streaming kernels (HBM-bound) over signal-sized tensors.
#include "bpm_common.h"
#include "../../include/bpmult_hip.h"

namespace {

constexpr int FNT = 256;

// out[r, c .. c+3] for r < total_rows: row r = b * rows_per_batch + front + l with l < L takes x[b*sb + c*sc + l*sl], every
// other row zeros.  One thread = 4 channels of one row (C % 4 == 0).
template <typename CT>
__global__ __launch_bounds__(FNT) void signal_pack_kernel(const float* __restrict__ x, CT* __restrict__ out, int B, int C, int L, long sb, long sc, long sl,
                                                          int front, int rows_per_batch, long total_rows, int ld) {
    const long idx = (long)blockIdx.x * FNT + threadIdx.x;
    const int cq = C >> 2;
    if (idx >= total_rows * cq) return;
    const long row = idx / cq;
    const int c = (int)(idx % cq) * 4;
    const long b = row / rows_per_batch;
    const int l = (int)(row % rows_per_batch) - front;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (b < B && l >= 0 && l < L) {
        const float* p = x + b * sb + c * sc + (long)l * sl;
        v0 = p[0]; v1 = p[sc]; v2 = p[2 * sc]; v3 = p[3 * sc];
    }
    CT* o = out + row * ld + c;
    if constexpr (sizeof(CT) == 4) *(f32x4*)o = f32x4{v0, v1, v2, v3};
    else { bf16x4 t; t[0] = (bf16_t)v0; t[1] = (bf16_t)v1; t[2] = (bf16_t)v2; t[3] = (bf16_t)v3; *(bf16x4*)o = t; }
}

// x[b*sb + c*sc + l*sl] = l < Lvalid ? src[(b * rows_per_batch + l) * ld + c] : 0     for l < L
__global__ __launch_bounds__(FNT) void signal_unpack_kernel(const float* __restrict__ src, float* __restrict__ x, int B, int C, int L, long sb, long sc, long sl,
                                                            int Lvalid, int rows_per_batch, int ld) {
    const long idx = (long)blockIdx.x * FNT + threadIdx.x;
    const int cq = C >> 2;
    if (idx >= (long)B * L * cq) return;
    const int c = (int)(idx % cq) * 4, l = (int)((idx / cq) % L);
    const long b = idx / ((long)cq * L);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (l < Lvalid) v = *(const f32x4*)(src + (b * rows_per_batch + l) * ld + c);
    float* p = x + b * sb + c * sc + (long)l * sl;
    p[0] = v[0]; p[sc] = v[1]; p[2 * sc] = v[2]; p[3 * sc] = v[3];
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

extern "C" int bpm_signal_pack(int dtype, const float* x, void* out, int B, int C, int L, int64_t sb, int64_t sc, int64_t sl,
                               int front, int rows_per_batch, int64_t total_rows, int ld, void* stream) {
    if (!x || !out || B < 1 || C < 4 || (C & 3) || L < 1 || front < 0 || rows_per_batch < front + L || ld < C) return BPM_ERR_ARG;
    if (total_rows < (int64_t)B * rows_per_batch) return BPM_ERR_ARG;
    if (dtype != BPM_F32 && dtype != BPM_BF16) return BPM_ERR_ARG;
    if (((uintptr_t)out & 15) || (ld & 3)) return BPM_ERR_ALIGN;
    const long n = (long)total_rows * (C >> 2);
    const dim3 grid((unsigned)((n + FNT - 1) / FNT));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == BPM_BF16) hipLaunchKernelGGL(signal_pack_kernel<bf16_t>, grid, dim3(FNT), 0, s, x, (bf16_t*)out, B, C, L, sb, sc, sl, front, rows_per_batch, total_rows, ld);
    else hipLaunchKernelGGL(signal_pack_kernel<float>, grid, dim3(FNT), 0, s, x, (float*)out, B, C, L, sb, sc, sl, front, rows_per_batch, total_rows, ld);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_signal_unpack(const float* src, float* x, int B, int C, int L, int64_t sb, int64_t sc, int64_t sl,
                                 int Lvalid, int rows_per_batch, int ld, void* stream) {
    if (!src || !x || B < 1 || C < 4 || (C & 3) || L < 1 || Lvalid < 0 || Lvalid > L || rows_per_batch < Lvalid || ld < C) return BPM_ERR_ARG;
    if (((uintptr_t)src & 15) || (ld & 3)) return BPM_ERR_ALIGN;
    const long n = (long)B * L * (C >> 2);
    hipLaunchKernelGGL(signal_unpack_kernel, dim3((unsigned)((n + FNT - 1) / FNT)), dim3(FNT), 0, (hipStream_t)stream, src, x, B, C, L, sb, sc, sl,
                       Lvalid, rows_per_batch, ld);
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
