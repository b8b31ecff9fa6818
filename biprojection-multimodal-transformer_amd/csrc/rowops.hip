// HBM-bound row kernels of the BPMulT hot path (gfx950): input staging,
// weight shadows, embedding scale + positional term, LayerNorm, gradient
// casts with bias reduction, Fusion-GMU gating.  Each is a streaming kernel
// whose roofline is HBM bandwidth; rows are [(t*B + b), d] fp32 on the
// residual stream and CT (f32 / bf16, leading dim padded to 32 with zeros)
// where the consumer is an MFMA GEMM.
#include "bpm_common.h"
#include "../../include/bpmult_hip.h"

namespace {

constexpr int NT = 256;

template <typename CT> BPM_DEV void put(void* p, size_t i, float v) { ((CT*)p)[i] = Tr<CT>::from_f(v); }

inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) {
    DropCfg d;
    d.thresh = 0; d.key = 0; d.inv_keep = 1.f;
    if (p > 0.f) {
        d.thresh = (uint32_t)(p * 16777216.0 + 0.5);
        d.key = bpm_host_drop_key(seed, site);
        d.inv_keep = 1.f / (1.f - p);
    }
    return d;
}

inline int grid_for(size_t n, int per_block) {
    size_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > 65535u * 16u) g = 65535u * 16u;
    return (int)g;
}

// ---------------------------------------------------------------------------
// input staging: src fp32 [B,T,C]  <->  packed CT [(t*B+b), ld]
// (reference mmtr.py:741-753: dropout on the raw text features, transpose,
//  permute(2,0,1); the permutation is done here on the read side)
// ---------------------------------------------------------------------------
template <typename CT>
__global__ void pack_rows_fwd_kernel(const float* __restrict__ src, void* dst, int B, int T, int C, int ld, DropCfg drop) {
    const size_t total = (size_t)T * B * ld;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % ld);
        const size_t row = i / ld;
        const int b = (int)(row % B), t = (int)(row / B);
        float v = 0.f;
        if (c < C) {
            const size_t si = ((size_t)b * T + t) * C + c;
            v = src[si] * bpm_drop_mult(drop, (uint32_t)si);
        }
        put<CT>(dst, i, v);
    }
}

// d(src)[b,t,c] = drop_mult * g[(t*B+b), c]   (g fp32, leading dim ldg)
__global__ void pack_rows_bwd_kernel(const float* __restrict__ g, int ldg, float* dsrc, int B, int T, int C, DropCfg drop) {
    const size_t total = (size_t)T * B * C;
    for (size_t si = (size_t)blockIdx.x * NT + threadIdx.x; si < total; si += (size_t)gridDim.x * NT) {
        const int c = (int)(si % C);
        const size_t bt = si / C;
        const int t = (int)(bt % T), b = (int)(bt / T);
        dsrc[si] = g[((size_t)t * B + b) * ldg + c] * bpm_drop_mult(drop, (uint32_t)si);
    }
}

// ---------------------------------------------------------------------------
// weight shadows: every fp32 master [rows, cols] -> CT [rows, ld] (zero pad),
// one launch over a device-resident table
// ---------------------------------------------------------------------------
template <typename CT>
__global__ void pack_weights_kernel(const bpm_pack_desc* __restrict__ tab, int ndesc) {
    // block -> descriptor by binary search on blk0
    int lo = 0, hi = ndesc - 1;
    const unsigned bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
    }
    const bpm_pack_desc d = tab[lo];
    const size_t total = (size_t)d.rows * d.ld;
    const size_t i0 = ((size_t)(bid - d.blk0) * NT + threadIdx.x) * 4;
    const float* src = (const float*)d.src;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t i = i0 + e;
        if (i >= total) break;
        const int c = (int)(i % d.ld);
        const size_t r = i / d.ld;
        put<CT>(d.dst, i, c < d.cols ? src[r * d.cols + c] : 0.f);
    }
}

// ---------------------------------------------------------------------------
// embedding prologue (reference transformer.py:66-79, position_embedding.py:62-76)
//   out = dropout(scale * x + table[pos]),  pos = t+1 if x[t,b,0] != 0 else 0
// ---------------------------------------------------------------------------
__global__ void embed_pos_fwd_kernel(const float* __restrict__ x, const float* __restrict__ table, float* out,
                                     int T, int B, int d, float scale, DropCfg drop) {
    const size_t total = (size_t)T * B * d;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % d);
        const size_t row = i / d;
        const int t = (int)(row / B);
        const int pos = (x[row * d] != 0.f) ? t + 1 : 0;
        out[i] = (scale * x[i] + table[(size_t)pos * d + c]) * bpm_drop_mult(drop, (uint32_t)i);
    }
}

// dx (+)= scale * drop_mult * dy     (the positional term is detached)
__global__ void embed_pos_bwd_kernel(const float* __restrict__ dy, float* dx, size_t total, float scale, DropCfg drop, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const float v = scale * dy[i] * bpm_drop_mult(drop, (uint32_t)i);
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// ---------------------------------------------------------------------------
// LayerNorm, eps inside the sqrt, biased variance (nn.LayerNorm).  One wave
// per row, row held in registers (d <= 64*MAXE).
// ---------------------------------------------------------------------------
constexpr int MAXE = 32;

template <typename CT, bool F32OUT, int NE>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, void* out, int ldo,
                                                    float* mean, float* rstd, int R, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int wpb = NT / 64;
    for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < R; row += gridDim.x * wpb) {
        const float* xr = x + (size_t)row * d;
        float v[NE];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            v[e] = c < d ? xr[c] : 0.f;
            s += v[e];
        }
        const float mu = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            const float t = c < d ? v[e] - mu : 0.f;
            q += t * t;
        }
        const float rs = rsqrtf(wave_sum(q) / d + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            if (c < d) {
                const float y = (v[e] - mu) * rs * gamma[c] + beta[c];
                if (F32OUT) ((float*)out)[(size_t)row * ldo + c] = y;
                else put<CT>(out, (size_t)row * ldo + c, y);
            } else if (!F32OUT && c < ldo) {
                put<CT>(out, (size_t)row * ldo + c, 0.f);
            }
        }
    }
}

// dx = add + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// dgamma += sum_rows dy * xhat,  dbeta += sum_rows dy   (atomics, one per column per block)
template <int NE>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* add, float* dx,
                                                    float* dgamma, float* dbeta, int R, int d) {
    __shared__ float red[2][NT / 64][512];   // 512 columns per pass
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wpb = NT / 64;
    float ag[NE], ab[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { ag[e] = 0.f; ab[e] = 0.f; }
    for (int row = blockIdx.x * wpb + wv; row < R; row += gridDim.x * wpb) {
        const float mu = mean[row], rs = rstd[row];
        const float* xr = x + (size_t)row * d;
        const float* gr = dy + (size_t)row * ldy;
        float xh[NE], gg[NE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            if (c < d) {
                const float dyv = gr[c];
                xh[e] = (xr[c] - mu) * rs;
                gg[e] = dyv * gamma[c];
                ag[e] += dyv * xh[e];
                ab[e] += dyv;
                s1 += gg[e];
                s2 += gg[e] * xh[e];
            } else { xh[e] = 0.f; gg[e] = 0.f; }
        }
        s1 = wave_sum(s1) / d;
        s2 = wave_sum(s2) / d;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            if (c < d) {
                const float v = rs * (gg[e] - s1 - xh[e] * s2);
                const size_t i = (size_t)row * d + c;
                dx[i] = add ? add[i] + v : v;
            }
        }
    }
    if (!dgamma) return;
    // reduce the 4 waves' partial column sums through LDS, 8 register slots (512 columns) at a time
#pragma unroll
    for (int e0 = 0; e0 < NE; e0 += 8) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[0][wv][lane + 64 * e] = (e0 + e < NE) ? ag[(e0 + e < NE) ? e0 + e : 0] : 0.f;
            red[1][wv][lane + 64 * e] = (e0 + e < NE) ? ab[(e0 + e < NE) ? e0 + e : 0] : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 512; i += NT) {
            const int c = e0 * 64 + i;
            if (c < d) {
                float sg = 0.f, sb = 0.f;
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) { sg += red[0][w][i]; sb += red[1][w][i]; }
                atomicAdd(dgamma + c, sg);
                atomicAdd(dbeta + c, sb);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// rows_cast: y = (a [+ b]) * drop_mult ; written as CT (padded) and/or fp32;
// optional column sums (bias gradient) by atomics.
// ---------------------------------------------------------------------------
constexpr int CAST_ROWS = 32;

template <typename CT>
__global__ __launch_bounds__(NT) void rows_cast_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                       void* dct, int ldd, float* df32, int ldf, float* colsum,
                                                       int R, int C, DropCfg drop) {
    const int c = blockIdx.y * NT + threadIdx.x;
    const int r0 = blockIdx.x * CAST_ROWS;
    const int r1 = min(R, r0 + CAST_ROWS);
    const int cmax = dct ? ldd : C;
    if (c >= cmax) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
        float v = 0.f;
        if (c < C) {
            v = a[(size_t)r * lda + c];
            if (b) v += b[(size_t)r * ldb + c];
            v *= bpm_drop_mult(drop, (uint32_t)r * (uint32_t)C + (uint32_t)c);
            if (df32) df32[(size_t)r * ldf + c] = v;
            s += v;
        }
        if (dct) put<CT>(dct, (size_t)r * ldd + c, v);
    }
    if (colsum && c < C) atomicAdd(colsum + c, s);
}

// ---------------------------------------------------------------------------
// Fusion-GMU gating (reference mmtr.py:189-195)
//   out = z*tanh(a1)*x1 + (1-z)*tanh(a2)*x2,  z = sigmoid(ag)
// a1,a2,ag are the three bias-free linear maps produced by the GEMM kernel.
// ---------------------------------------------------------------------------
BPM_DEV float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ void gmu2_fwd_kernel(const float* __restrict__ a1, const float* __restrict__ a2, const float* __restrict__ ag,
                                const float* __restrict__ x1, const float* __restrict__ x2, float* out, size_t total) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const float z = sigmoidf_(ag[i]);
        out[i] = z * tanhf(a1[i]) * x1[i] + (1.f - z) * tanhf(a2[i]) * x2[i];
    }
}

// da1, da2, dag -> CT [R, ldg] (GEMM operands, pad zeroed); dx1, dx2 (direct terms) -> fp32 [R, d]
template <typename CT>
__global__ void gmu2_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a1, const float* __restrict__ a2,
                                const float* __restrict__ ag, const float* __restrict__ x1, const float* __restrict__ x2,
                                void* da1, void* da2, void* dag, int ldg, float* dx1, float* dx2, int R, int d) {
    const size_t total = (size_t)R * ldg;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % ldg);
        const size_t r = i / ldg;
        float g1 = 0.f, g2 = 0.f, gz = 0.f;
        if (c < d) {
            const size_t k = r * d + c;
            const float go = dout[k];
            const float z = sigmoidf_(ag[k]);
            const float h1 = tanhf(a1[k]), h2 = tanhf(a2[k]);
            const float u1 = x1[k], u2 = x2[k];
            g1 = go * z * u1 * (1.f - h1 * h1);
            g2 = go * (1.f - z) * u2 * (1.f - h2 * h2);
            gz = go * (h1 * u1 - h2 * u2) * z * (1.f - z);
            dx1[k] = go * z * h1;
            dx2[k] = go * (1.f - z) * h2;
        }
        put<CT>(da1, i, g1);
        put<CT>(da2, i, g2);
        put<CT>(dag, i, gz);
    }
}

}  // namespace

#define BPM_DISPATCH_CT(dtype, EXPR_F32, EXPR_BF16)   \
    do {                                              \
        if ((dtype) == BPM_BF16) { EXPR_BF16; }       \
        else { EXPR_F32; }                            \
    } while (0)

extern "C" int bpm_pack_rows_fwd(int dtype, const float* src, void* dst, int B, int T, int C, int ld,
                                 float drop_p, uint64_t seed, uint32_t site, void* stream) {
    if (!src || !dst || B < 1 || T < 1 || C < 1 || ld < C) return BPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const DropCfg dr = make_drop(drop_p, seed, site);
    const int grid = grid_for((size_t)T * B * ld, NT * 4);
    BPM_DISPATCH_CT(dtype,
        hipLaunchKernelGGL(pack_rows_fwd_kernel<float>, dim3(grid), dim3(NT), 0, s, src, dst, B, T, C, ld, dr),
        hipLaunchKernelGGL(pack_rows_fwd_kernel<bf16_t>, dim3(grid), dim3(NT), 0, s, src, dst, B, T, C, ld, dr));
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_pack_rows_bwd(const float* g, int ldg, float* dsrc, int B, int T, int C,
                                 float drop_p, uint64_t seed, uint32_t site, void* stream) {
    if (!g || !dsrc || B < 1 || T < 1 || C < 1 || ldg < C) return BPM_ERR_ARG;
    const DropCfg dr = make_drop(drop_p, seed, site);
    hipLaunchKernelGGL(pack_rows_bwd_kernel, dim3(grid_for((size_t)T * B * C, NT * 4)), dim3(NT), 0, (hipStream_t)stream,
                       g, ldg, dsrc, B, T, C, dr);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_pack_weights(int dtype, const bpm_pack_desc* table_dev, int ndesc, unsigned total_blocks, void* stream) {
    if (!table_dev || ndesc < 1 || total_blocks < 1) return BPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    BPM_DISPATCH_CT(dtype,
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(total_blocks), dim3(NT), 0, s, table_dev, ndesc),
        hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(total_blocks), dim3(NT), 0, s, table_dev, ndesc));
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_embed_pos_fwd(const float* x, const float* table, int table_rows, float* out, int T, int B, int d,
                                 float scale, float drop_p, uint64_t seed, uint32_t site, void* stream) {
    if (!x || !table || !out || T < 1 || B < 1 || d < 1 || table_rows < T + 1) return BPM_ERR_ARG;
    const DropCfg dr = make_drop(drop_p, seed, site);
    hipLaunchKernelGGL(embed_pos_fwd_kernel, dim3(grid_for((size_t)T * B * d, NT * 4)), dim3(NT), 0, (hipStream_t)stream,
                       x, table, out, T, B, d, scale, dr);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_embed_pos_bwd(const float* dy, float* dx, int T, int B, int d, float scale, float drop_p,
                                 uint64_t seed, uint32_t site, int accumulate, void* stream) {
    if (!dy || !dx || T < 1 || B < 1 || d < 1) return BPM_ERR_ARG;
    const DropCfg dr = make_drop(drop_p, seed, site);
    const size_t total = (size_t)T * B * d;
    hipLaunchKernelGGL(embed_pos_bwd_kernel, dim3(grid_for(total, NT * 4)), dim3(NT), 0, (hipStream_t)stream,
                       dy, dx, total, scale, dr, accumulate);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_ln_fwd(int out_dtype, const float* x, const float* gamma, const float* beta, void* out, int ldo,
                          float* mean, float* rstd, int R, int d, float eps, void* stream) {
    if (!x || !gamma || !beta || !out || !mean || !rstd || R < 1 || d < 1 || d > 64 * MAXE || ldo < d || ldo > 64 * MAXE) return BPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int grid = min(4096, (R + 3) / 4);
    // the padded row (ldo columns for CT outputs) must fit the lane-strided register tile
    const int span = out_dtype == BPM_OUT_LN_F32 ? d : ldo;
#define BPM_LN_FWD(NE)                                                                                                              \
    if (span <= 64 * NE) {                                                                                                          \
        if (out_dtype == BPM_OUT_LN_F32)                                                                                            \
            hipLaunchKernelGGL((ln_fwd_kernel<float, true, NE>), dim3(grid), dim3(NT), 0, s, x, gamma, beta, out, ldo, mean, rstd, R, d, eps); \
        else if (out_dtype == BPM_BF16)                                                                                             \
            hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, false, NE>), dim3(grid), dim3(NT), 0, s, x, gamma, beta, out, ldo, mean, rstd, R, d, eps); \
        else                                                                                                                        \
            hipLaunchKernelGGL((ln_fwd_kernel<float, false, NE>), dim3(grid), dim3(NT), 0, s, x, gamma, beta, out, ldo, mean, rstd, R, d, eps); \
        BPM_CHECK_LAUNCH();                                                                                                         \
        return 0;                                                                                                                   \
    }
    BPM_LN_FWD(1) BPM_LN_FWD(2) BPM_LN_FWD(5) BPM_LN_FWD(8) BPM_LN_FWD(12) BPM_LN_FWD(16) BPM_LN_FWD(24) BPM_LN_FWD(32)
#undef BPM_LN_FWD
    return BPM_ERR_ARG;
}

extern "C" int bpm_ln_bwd(const float* dy, int ldy, const float* x, const float* mean, const float* rstd, const float* gamma,
                          const float* add, float* dx, float* dgamma, float* dbeta, int R, int d, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || R < 1 || d < 1 || d > 64 * MAXE || ldy < d) return BPM_ERR_ARG;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return BPM_ERR_ARG;
    const int grid = min(256, (R + 3) / 4);
#define BPM_LN_BWD(NE)                                                                                                   \
    if (d <= 64 * NE) {                                                                                                  \
        hipLaunchKernelGGL(ln_bwd_kernel<NE>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, dy, ldy, x, mean, rstd, gamma, \
                           add, dx, dgamma, dbeta, R, d);                                                                \
        BPM_CHECK_LAUNCH();                                                                                              \
        return 0;                                                                                                        \
    }
    BPM_LN_BWD(1) BPM_LN_BWD(2) BPM_LN_BWD(5) BPM_LN_BWD(8) BPM_LN_BWD(12) BPM_LN_BWD(16) BPM_LN_BWD(24) BPM_LN_BWD(32)
#undef BPM_LN_BWD
    return BPM_ERR_ARG;
}

extern "C" int bpm_rows_cast(int dtype, const float* a, int lda, const float* b, int ldb, void* dst_ct, int ldd,
                             float* dst_f32, int ldf, float* colsum, int R, int C,
                             float drop_p, uint64_t seed, uint32_t site, void* stream) {
    if (!a || R < 1 || C < 1 || lda < C || (b && ldb < C) || (dst_ct && ldd < C) || (dst_f32 && ldf < C)) return BPM_ERR_ARG;
    if (!dst_ct && !dst_f32 && !colsum) return BPM_ERR_ARG;
    const DropCfg dr = make_drop(drop_p, seed, site);
    const int cols = dst_ct ? ldd : C;
    dim3 grid((R + CAST_ROWS - 1) / CAST_ROWS, (cols + NT - 1) / NT);
    hipStream_t s = (hipStream_t)stream;
    BPM_DISPATCH_CT(dtype,
        hipLaunchKernelGGL(rows_cast_kernel<float>, grid, dim3(NT), 0, s, a, lda, b, ldb, dst_ct, ldd, dst_f32, ldf, colsum, R, C, dr),
        hipLaunchKernelGGL(rows_cast_kernel<bf16_t>, grid, dim3(NT), 0, s, a, lda, b, ldb, dst_ct, ldd, dst_f32, ldf, colsum, R, C, dr));
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_gmu2_fwd(const float* a1, const float* a2, const float* ag, const float* x1, const float* x2,
                            float* out, int R, int d, void* stream) {
    if (!a1 || !a2 || !ag || !x1 || !x2 || !out || R < 1 || d < 1) return BPM_ERR_ARG;
    const size_t total = (size_t)R * d;
    hipLaunchKernelGGL(gmu2_fwd_kernel, dim3(grid_for(total, NT * 4)), dim3(NT), 0, (hipStream_t)stream, a1, a2, ag, x1, x2, out, total);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_gmu2_bwd(int dtype, const float* dout, const float* a1, const float* a2, const float* ag,
                            const float* x1, const float* x2, void* da1, void* da2, void* dag, int ldg,
                            float* dx1, float* dx2, int R, int d, void* stream) {
    if (!dout || !a1 || !a2 || !ag || !x1 || !x2 || !da1 || !da2 || !dag || !dx1 || !dx2 || R < 1 || d < 1 || ldg < d) return BPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int grid = grid_for((size_t)R * ldg, NT * 4);
    BPM_DISPATCH_CT(dtype,
        hipLaunchKernelGGL(gmu2_bwd_kernel<float>, dim3(grid), dim3(NT), 0, s, dout, a1, a2, ag, x1, x2, da1, da2, dag, ldg, dx1, dx2, R, d),
        hipLaunchKernelGGL(gmu2_bwd_kernel<bf16_t>, dim3(grid), dim3(NT), 0, s, dout, a1, a2, ag, x1, x2, da1, da2, dag, ldg, dx1, dx2, R, d));
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_version(void) { return BPM_ABI_VERSION; }

extern "C" const char* bpm_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case BPM_ERR_ARG: return "bpmult_hip: invalid argument (shape / pointer / mode)";
        case BPM_ERR_ALIGN: return "bpmult_hip: operand not 16-byte aligned (pointer or leading dimension)";
        default: return hipGetErrorString((hipError_t)code);
    }
}
