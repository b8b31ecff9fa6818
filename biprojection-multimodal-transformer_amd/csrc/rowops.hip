// HBM-bound row kernels of the BPMulT hot path (gfx950): input staging,
// weight shadows, embedding scale + positional term, LayerNorm, gradient
// casts with bias reduction, Fusion-GMU gating.  Each is a streaming kernel
// whose roofline is HBM bandwidth; rows are [(t*B + b), d] fp32 on the
// residual stream and CT (f32 / bf16, leading dim padded to 32 with zeros)
// where the consumer is an MFMA GEMM.
//
// Every entry point is GROUPED: one launch serves up to BPM_MAX_GROUP
// independent problems (the encoders of one level run in lock-step), passed
// by value in the kernel-argument segment; a block finds its problem from a
// prefix table of block counts.
#include <cmath>

#include "bpm_common.h"
#include "../../include/bpmult_hip.h"

constexpr int BPM_BASE_PRIO = 1;      // see gemm.hip

namespace {

constexpr int NT = 256;

template <typename CT> BPM_DEV void put(void* p, size_t i, float v) { ((CT*)p)[i] = Tr<CT>::from_f(v); }
template <typename CT>
BPM_DEV void put4(void* p, size_t i, f32x4 v) {
    if constexpr (sizeof(CT) == 4) *(f32x4*)((float*)p + i) = v;
    else { bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3]; *(bf16x4*)((bf16_t*)p + i) = o; }
}

inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) { return bpm_make_drop(p, seed, site); }

inline unsigned blocks_for(size_t n, int per_block, unsigned cap) {
    size_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

template <typename P> struct Grp {
    const uint64_t* seedp;              // dropout seed read at execution time (BPM_SEED_INDIRECT), or nullptr
    int n;
    unsigned blk0[BPM_MAX_GROUP + 1];   // prefix of block counts
    P p[BPM_MAX_GROUP];
};

// block -> (problem index, block within problem, blocks of that problem)
template <typename P>
BPM_DEV const P& pick(const Grp<P>& g, unsigned& bid, unsigned& nblk) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if (bid >= g.blk0[i]) pi = i;
    nblk = g.blk0[pi + 1] - g.blk0[pi];
    bid -= g.blk0[pi];
    return g.p[pi];
}

// ---------------------------------------------------------------------------
// input staging: src fp32 [B,T,C]  <->  packed CT [(t*B+b), ld]
// (reference mmtr.py:741-753: dropout on the raw text features, transpose,
//  permute(2,0,1); the permutation is done here on the read side)
// ---------------------------------------------------------------------------
struct PackP { const float* src; void* dst; float* dsrc; const float* g; int ldg; int B, T, C, ld; DropCfg drop; };

template <typename CT>
__global__ void pack_rows_fwd_kernel(const Grp<PackP> grp) {
    unsigned bid = blockIdx.x, nblk;
    const PackP& P = pick(grp, bid, nblk);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const size_t total = (size_t)P.T * P.B * P.ld;
    for (size_t i = (size_t)bid * NT + threadIdx.x; i < total; i += (size_t)nblk * NT) {
        const int c = (int)(i % P.ld);
        const size_t row = i / P.ld;
        const int b = (int)(row % P.B), t = (int)(row / P.B);
        float v = 0.f;
        if (c < P.C) {
            const size_t si = ((size_t)b * P.T + t) * P.C + c;
            v = P.src[si] * bpm_drop_mult(drop, (uint32_t)si);
        }
        put<CT>(P.dst, i, v);
    }
}

// d(src)[b,t,c] = drop_mult * g[(t*B+b), c]   (g fp32, leading dim ldg)
__global__ void pack_rows_bwd_kernel(const Grp<PackP> grp) {
    unsigned bid = blockIdx.x, nblk;
    const PackP& P = pick(grp, bid, nblk);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const size_t total = (size_t)P.T * P.B * P.C;
    for (size_t si = (size_t)bid * NT + threadIdx.x; si < total; si += (size_t)nblk * NT) {
        const int c = (int)(si % P.C);
        const size_t bt = si / P.C;
        const int t = (int)(bt % P.T), b = (int)(bt / P.T);
        P.dsrc[si] = P.g[((size_t)t * P.B + b) * P.ldg + c] * bpm_drop_mult(drop, (uint32_t)si);
    }
}

// ---------------------------------------------------------------------------
// weight shadows: every fp32 master [rows, cols] (row stride src_ld) -> CT
// [rows, ld] (zero pad), one launch over a device-resident table
// ---------------------------------------------------------------------------
template <typename CT>
__global__ void pack_weights_kernel(const bpm_pack_desc* __restrict__ tab, int ndesc) {
    int lo = 0, hi = ndesc - 1;
    const unsigned bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
    }
    const bpm_pack_desc d = tab[lo];
    const size_t total = (size_t)d.rows * d.ld;
    const size_t i0 = ((size_t)(bid - d.blk0) * NT + threadIdx.x) * 4;
    if (i0 >= total) return;
    const float* src = (const float*)d.src;
    // ld % 4 == 0: the thread's 4 elements share a row
    const size_t r = i0 / d.ld;
    const int c = (int)(i0 - r * d.ld);
    const float* sp = src + r * d.src_ld + c;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((d.ld & 3) == 0 && c + 3 < d.cols && (((uintptr_t)sp) & 15) == 0) {
        v = *(const f32x4*)sp;
        if (d.colscale) v *= *(const f32x4*)(d.colscale + c);     // colscale is 64-byte aligned and c % 4 == 0
        if ((((uintptr_t)d.dst) & 15) == 0 && (d.dst_ld & 3) == 0) { put4<CT>(d.dst, r * d.dst_ld + c, v); return; }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const size_t i = i0 + e;
            if (i >= total) break;
            const int ce = (int)(i % d.ld);
            const size_t re = i / d.ld;
            float x = 0.f;
            if (ce < d.cols) {
                x = src[re * d.src_ld + ce];
                if (d.colscale) x *= d.colscale[ce];
            }
            put<CT>(d.dst, re * d.dst_ld + ce, x);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) put<CT>(d.dst, r * d.dst_ld + c + e, v[e]);
}

template <typename D>
BPM_DEV const D& find_desc(const D* __restrict__ tab, int ndesc, unsigned bid) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
    }
    return tab[lo];
}

// folded bias: out[n] = b[n] + W[n,:] . beta   (one wave per row)
__global__ __launch_bounds__(NT) void fold_bias_kernel(const bpm_fold_desc* __restrict__ tab, int ndesc) {
    const bpm_fold_desc d = find_desc(tab, ndesc, blockIdx.x);
    const int row = (int)(blockIdx.x - d.blk0) * (NT / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= d.rows) return;
    float s = 0.f;
    for (int c = lane; c < d.cols; c += 64) s += d.W[(size_t)row * d.ldw + c] * d.beta[c];
    s = wave_sum(s);
    if (lane == 0) d.out[row] = s + (d.b ? d.b[row] : 0.f);
}

// gradients of the real parameters from the folded ones (see bpm_unfold_desc)
// (Measured in round 3 and not kept: per-block partial rows through a workspace + last-block sum instead of the float
// atomics on dgamma / dbeta -- 72 us against 42 stand-alone at hidden 768: the 4-byte write-through stores cost more
// than the contended atomics.)
constexpr int UNF_ROWS = 16;
__global__ __launch_bounds__(NT) void unfold_grads_kernel(const bpm_unfold_desc* __restrict__ tab, int ndesc, int store_dw) {
    const bpm_unfold_desc d = find_desc(tab, ndesc, blockIdx.x);
    const int r0 = (int)(blockIdx.x - d.blk0) * UNF_ROWS, r1 = min(d.rows, r0 + UNF_ROWS);
    for (int c = threadIdx.x; c < d.cols; c += NT) {
        const float g = d.gamma[c], bt = d.beta[c];
        float ag = 0.f, ab = 0.f;
        for (int r = r0; r < r1; ++r) {
            const float f = d.dWf[(size_t)r * d.cols + c], w = d.W[(size_t)r * d.ldw + c], db = d.dbf[r];
            const float v = f * g + db * bt;
            float* o = d.dW + (size_t)r * d.ldw + c;
            *o = store_dw ? v : *o + v;                   // store_dw: this launch is the first writer of dW this step
            ag += f * w;
            ab += db * w;
        }
        atomicAdd(d.dgamma + c, ag);
        atomicAdd(d.dbeta + c, ab);
    }
    if ((int)threadIdx.x < r1 - r0) d.dbias[r0 + threadIdx.x] += d.dbf[r0 + threadIdx.x];
}

// zero a list of fp32 segments (device-resident table): the small tensors of a flat gradient buffer whose large ones are
// written by plain stores (see bpm_zero_segments)
constexpr unsigned ZSEG = NT * 16;          // elements per block
__global__ __launch_bounds__(NT) void zero_segments_kernel(const bpm_zero_desc* __restrict__ tab, int ndesc) {
    const bpm_zero_desc d = find_desc(tab, ndesc, blockIdx.x);
    const unsigned i0 = (blockIdx.x - d.blk0) * ZSEG, i1 = min(d.n, i0 + ZSEG);
    if ((((uintptr_t)d.p) & 15) == 0) {
        for (unsigned i = i0 + 4 * threadIdx.x; i + 3 < i1; i += 4 * NT) *(f32x4*)(d.p + i) = f32x4{0.f, 0.f, 0.f, 0.f};
        for (unsigned i = i0 + ((i1 - i0) & ~3u) + threadIdx.x; i < i1; i += NT) d.p[i] = 0.f;
    } else {
        for (unsigned i = i0 + threadIdx.x; i < i1; i += NT) d.p[i] = 0.f;
    }
}

// ---------------------------------------------------------------------------
// embedding prologue (reference transformer.py:66-79, position_embedding.py:62-76)
//   out = dropout(scale * x + table[pos]),  pos = t+1 if x[t,b,0] != 0 else 0
// backward: dx (+)= scale * drop_mult * dy   (the positional term is detached)
// ---------------------------------------------------------------------------
struct EmbP { const float* x; float* out; int T, B; int accumulate; int pos0, pstride; DropCfg drop; };

// four consecutive channels per lane (d % 4 == 0, 16-byte aligned tensors, < 2^32 elements): 32-bit index arithmetic, one
// dropout hash per quad, 16-byte accesses; the element-wise loops below are the general path
BPM_DEV bool emb_wide(const EmbP& P, const float* table, int d, size_t total) {
    return (d & 3) == 0 && total < (1ull << 32) && (((uintptr_t)P.x | (uintptr_t)P.out | (uintptr_t)table) & 15) == 0;
}

__global__ void embed_pos_fwd_kernel(const Grp<EmbP> grp, const float* __restrict__ table, int d, float scale) {
    unsigned bid = blockIdx.x, nblk;
    const EmbP& P = pick(grp, bid, nblk);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const size_t total = (size_t)P.T * P.B * d;
    if (emb_wide(P, table, d, total)) {
        const uint32_t total4 = (uint32_t)(total >> 2), ud = (uint32_t)d;
        for (uint32_t q = bid * NT + threadIdx.x; q < total4; q += nblk * NT) {
            const uint32_t i = 4u * q, row = i / ud, c = i - row * ud;
            const int t = (int)(row / (uint32_t)P.B);
            const int pos = (P.x[(size_t)row * ud] != 0.f) ? P.pos0 + t * P.pstride + 1 : 0;
            f32x4 v = scale * *(const f32x4*)(P.x + i) + *(const f32x4*)(table + (size_t)pos * ud + c);
            if (drop.thresh != 0) {
                float d0, d1, d2, d3;
                bpm_drop_mult4(drop, i, d0, d1, d2, d3);
                v *= f32x4{d0, d1, d2, d3};
            }
            *(f32x4*)(P.out + i) = v;
        }
        return;
    }
    for (size_t i = (size_t)bid * NT + threadIdx.x; i < total; i += (size_t)nblk * NT) {
        const int c = (int)(i % d);
        const size_t row = i / d;
        const int t = (int)(row / P.B);
        const int pos = (P.x[row * d] != 0.f) ? P.pos0 + t * P.pstride + 1 : 0;     // row t sits at time pos0 + t*pstride
        P.out[i] = (scale * P.x[i] + table[(size_t)pos * d + c]) * bpm_drop_mult(drop, (uint32_t)i);
    }
}

__global__ void embed_pos_bwd_kernel(const Grp<EmbP> grp, int d, float scale) {
    unsigned bid = blockIdx.x, nblk;
    const EmbP& P = pick(grp, bid, nblk);      // x = dy, out = dx
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const size_t total = (size_t)P.T * P.B * d;
    if (emb_wide(P, nullptr, d, total)) {
        const uint32_t total4 = (uint32_t)(total >> 2);
        for (uint32_t q = bid * NT + threadIdx.x; q < total4; q += nblk * NT) {
            const uint32_t i = 4u * q;
            f32x4 v = scale * *(const f32x4*)(P.x + i);
            if (drop.thresh != 0) {
                float d0, d1, d2, d3;
                bpm_drop_mult4(drop, i, d0, d1, d2, d3);
                v *= f32x4{d0, d1, d2, d3};
            }
            f32x4* o = (f32x4*)(P.out + i);
            *o = P.accumulate ? *o + v : v;
        }
        return;
    }
    for (size_t i = (size_t)bid * NT + threadIdx.x; i < total; i += (size_t)nblk * NT) {
        const float v = scale * P.x[i] * bpm_drop_mult(drop, (uint32_t)i);
        P.out[i] = P.accumulate ? P.out[i] + v : v;
    }
}

// ---------------------------------------------------------------------------
// LayerNorm, eps inside the sqrt, biased variance (nn.LayerNorm).  One wave
// per row, row held in registers (NE = ceil(span/64) elements per lane).
// ---------------------------------------------------------------------------
constexpr int MAXE = 32;
constexpr int LN_WS_HEAD = 64;             // floats in front of the partial rows of a bpm_ln_bwd_ws workspace (ticket words)

struct LnP {
    const float* x; const float* gamma; const float* beta; void* out; int ldo; int out_f32;
    float* mean; float* rstd; int R;
    const float* dy; int ldy; const float* add; float* dx; float* dgamma; float* dbeta;
    void* cast; int ldc; float* csum; DropCfg drop;
};

template <typename CT, int NE>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(const Grp<LnP> grp, int d, float eps) {
    unsigned bid = blockIdx.x, nblk;
    const LnP& P = pick(grp, bid, nblk);
    const int lane = threadIdx.x & 63;
    const int wpb = NT / 64;
    for (int row = bid * wpb + (threadIdx.x >> 6); row < P.R; row += nblk * wpb) {
        const float* xr = P.x + (size_t)row * d;
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
        if (lane == 0) { P.mean[row] = mu; P.rstd[row] = rs; }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            if (c < d) {
                const float y = (v[e] - mu) * rs * P.gamma[c] + P.beta[c];
                if (P.out_f32) ((float*)P.out)[(size_t)row * P.ldo + c] = y;
                else put<CT>(P.out, (size_t)row * P.ldo + c, y);
            } else if (!P.out_f32 && c < P.ldo) {
                put<CT>(P.out, (size_t)row * P.ldo + c, 0.f);
            }
        }
    }
}

// dx = add + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// dgamma += sum_rows dy * xhat,  dbeta += sum_rows dy   (atomics, one per column per block)
// optional: cast[row, c] = CT(dx * dropout_mult(row*d + c)) (pad columns zeroed), csum += its column sums
template <typename CT, int NE>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const Grp<LnP> grp, int d) {
    __shared__ float red[3][NT / 64][512];   // 512 columns per pass
    unsigned bid = blockIdx.x, nblk;
    const LnP& P = pick(grp, bid, nblk);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wpb = NT / 64;
    const bool fused = P.cast != nullptr;          // uniform per block
    float ag[NE], ab[NE], ac[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { ag[e] = 0.f; ab[e] = 0.f; ac[e] = 0.f; }
    for (int row = bid * wpb + wv; row < P.R; row += nblk * wpb) {
        const float mu = P.mean[row], rs = P.rstd[row];
        const float* xr = P.x + (size_t)row * d;
        const float* gr = P.dy + (size_t)row * P.ldy;
        float xh[NE], gg[NE], av[NE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            if (c < d) {
                const float dyv = gr[c];
                av[e] = P.add ? P.add[(size_t)row * d + c] : 0.f;
                xh[e] = (xr[c] - mu) * rs;
                gg[e] = dyv * P.gamma[c];
                ag[e] += dyv * xh[e];
                ab[e] += dyv;
                s1 += gg[e];
                s2 += gg[e] * xh[e];
            } else { xh[e] = 0.f; gg[e] = 0.f; av[e] = 0.f; }
        }
        s1 = wave_sum(s1) / d;
        s2 = wave_sum(s2) / d;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            if (c < d) {
                const float v = av[e] + rs * (gg[e] - s1 - xh[e] * s2);
                P.dx[(size_t)row * d + c] = v;
                if (fused) {
                    const float m = v * bpm_drop_mult(drop, (uint32_t)row * (uint32_t)d + (uint32_t)c);
                    put<CT>(P.cast, (size_t)row * P.ldc + c, m);
                    ac[e] += m;
                }
            } else if (fused && c < P.ldc) {
                put<CT>(P.cast, (size_t)row * P.ldc + c, 0.f);
            }
        }
    }
    const bool want_g = P.dgamma != nullptr, want_c = fused && P.csum != nullptr;
    if (!want_g && !want_c) return;     // uniform per block
#pragma unroll
    for (int e0 = 0; e0 < NE; e0 += 8) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = e0 + e < NE;
            const int ei = in ? e0 + e : 0;
            red[0][wv][lane + 64 * e] = in ? ag[ei] : 0.f;
            red[1][wv][lane + 64 * e] = in ? ab[ei] : 0.f;
            red[2][wv][lane + 64 * e] = in ? ac[ei] : 0.f;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 512; i += NT) {
            const int c = e0 * 64 + i;
            if (c < d) {
                float sg = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
                for (int w = 0; w < NT / 64; ++w) { sg += red[0][w][i]; sb += red[1][w][i]; sc += red[2][w][i]; }
                if (want_g) { atomicAdd(P.dgamma + c, sg); atomicAdd(P.dbeta + c, sb); }
                if (want_c) atomicAdd(P.csum + c, sc);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Vector forms (d % 4 == 0, 16-byte aligned rows): a lane owns 4-column chunks lane, lane+64, ... (NV of them),
// moved as one 16-byte (f32) / 8-byte (bf16) access each; the backward keeps TWO rows in flight per wave.
// ---------------------------------------------------------------------------
BPM_DEV float sum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

template <typename CT, int NV>
__global__ __launch_bounds__(NT) void ln_fwd_vec_kernel(const Grp<LnP> grp, int d, float eps) {
    if (BPM_BASE_PRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    unsigned bid = blockIdx.x, nblk;
    const LnP& P = pick(grp, bid, nblk);
    const int lane = threadIdx.x & 63;
    const int wpb = NT / 64;
    const int nch = d >> 2;
    f32x4 gam[NV], bet[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int q = lane + 64 * e;
        const bool in = q < nch;
        gam[e] = *(const f32x4*)(P.gamma + 4 * (in ? q : 0));
        bet[e] = *(const f32x4*)(P.beta + 4 * (in ? q : 0));
    }
    const float inv_d = 1.f / d;
    for (int row = bid * wpb + (threadIdx.x >> 6); row < P.R; row += nblk * wpb) {
        const float* xr = P.x + (size_t)row * d;
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int q = lane + 64 * e;
            const f32x4 t = *(const f32x4*)(xr + 4 * (q < nch ? q : 0));
            v[e] = q < nch ? t : f32x4{0.f, 0.f, 0.f, 0.f};
            s += sum4(v[e]);
        }
        const float mu = wave_sum(s) * inv_d;
        float qq = 0.f;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int q = lane + 64 * e;
            f32x4 t = v[e] - mu;
            if (q >= nch) t = f32x4{0.f, 0.f, 0.f, 0.f};
            v[e] = t;
            qq += sum4(t * t);
        }
        const float rs = rsqrtf(wave_sum(qq) * inv_d + eps);
        if (lane == 0) { P.mean[row] = mu; P.rstd[row] = rs; }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const int q = lane + 64 * e;
            if (q < nch) {
                const f32x4 y = v[e] * rs * gam[e] + bet[e];
                if (P.out_f32) *(f32x4*)((float*)P.out + (size_t)row * P.ldo + 4 * q) = y;
                else put4<CT>(P.out, (size_t)row * P.ldo + 4 * q, y);
            } else if (!P.out_f32 && 4 * q < P.ldo) {
                put4<CT>(P.out, (size_t)row * P.ldo + 4 * q, f32x4{0.f, 0.f, 0.f, 0.f});
            }
        }
    }
}

template <typename CT, int NV>
__global__ __launch_bounds__(NT) void ln_bwd_vec_kernel(const Grp<LnP> grp, int d, float* __restrict__ ws) {
    if (BPM_BASE_PRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    __shared__ float red[3][NT / 64][NV * 256];
    unsigned bid = blockIdx.x, nblk;
    const LnP& P = pick(grp, bid, nblk);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wpb = NT / 64;
    const int nch = d >> 2;
    const bool fused = P.cast != nullptr;          // uniform per block
    const bool has_add = P.add != nullptr;
    const float inv_d = 1.f / d;
    f32x4 gam[NV], ag[NV], ab[NV], ac[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        const int q = lane + 64 * e;
        gam[e] = *(const f32x4*)(P.gamma + 4 * (q < nch ? q : 0));
        ag[e] = ab[e] = ac[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int U = 2;                             // rows in flight per wave
    for (int row0 = (bid * wpb + wv) * U; row0 < P.R; row0 += nblk * wpb * U) {
        f32x4 xv[U][NV], gv[U][NV], av[U][NV];
        float mu[U], rs[U];
        // phase 1: every load of the U rows (clamped rows / chunks, zeroed by selects)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = row0 + u;
            const int rc = row < P.R ? row : row0;
            mu[u] = P.mean[rc]; rs[u] = P.rstd[rc];
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const int q = lane + 64 * e;
                const int qc = q < nch ? q : 0;
                xv[u][e] = *(const f32x4*)(P.x + (size_t)rc * d + 4 * qc);
                gv[u][e] = *(const f32x4*)(P.dy + (size_t)rc * P.ldy + 4 * qc);
                av[u][e] = has_add ? *(const f32x4*)(P.add + (size_t)rc * d + 4 * qc) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = row0 + u;
            const bool rok = row < P.R;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const int q = lane + 64 * e;
                const bool ok = rok && q < nch;
                const f32x4 dyv = ok ? gv[u][e] : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 xh = ok ? (xv[u][e] - mu[u]) * rs[u] : f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 g = dyv * gam[e];
                ag[e] += dyv * xh;
                ab[e] += dyv;
                s1 += sum4(g);
                s2 += sum4(g * xh);
                xv[u][e] = xh; gv[u][e] = g;
            }
            s1 = wave_sum(s1) * inv_d;
            s2 = wave_sum(s2) * inv_d;
            if (!rok) continue;                      // wave-uniform
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const int q = lane + 64 * e;
                if (q < nch) {
                    const f32x4 v = av[u][e] + rs[u] * (gv[u][e] - s1 - xv[u][e] * s2);
                    *(f32x4*)(P.dx + (size_t)row * d + 4 * q) = v;
                    if (fused) {
                        f32x4 m = v;
                        if (drop.thresh != 0) {   // row*d + 4q is a multiple of 4 (d % 4 == 0): one hash quad
                            float d0, d1, d2, d3;
                            const uint32_t i0 = (uint32_t)row * (uint32_t)d + 4u * (uint32_t)q;
                            bpm_drop_mult4(drop, i0, d0, d1, d2, d3);
                            m[0] *= d0; m[1] *= d1; m[2] *= d2; m[3] *= d3;
                        }
                        put4<CT>(P.cast, (size_t)row * P.ldc + 4 * q, m);
                        ac[e] += m;
                    }
                } else if (fused && 4 * q < P.ldc) {
                    put4<CT>(P.cast, (size_t)row * P.ldc + 4 * q, f32x4{0.f, 0.f, 0.f, 0.f});
                }
            }
        }
    }
    const bool want_g = P.dgamma != nullptr, want_c = fused && P.csum != nullptr;
    if (!want_g && !want_c) return;                  // uniform per block
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        *(f32x4*)(&red[0][wv][4 * (lane + 64 * e)]) = ag[e];
        *(f32x4*)(&red[1][wv][4 * (lane + 64 * e)]) = ab[e];
        *(f32x4*)(&red[2][wv][4 * (lane + 64 * e)]) = ac[e];
    }
    __syncthreads();
    // Per-block column sums.  With a workspace every block stores its three rows and the LAST block of the problem to
    // finish adds them up (one owner per column, fixed order: bitwise reproducible) -- no second launch: a tiny dependent
    // kernel behind this one waited ~45 us for CU slots beside the side stream's GEMMs.  Without a workspace they go out
    // as float atomics -- up to 128 blocks per problem adding into the same three 3 KB rows, which runs an order of
    // magnitude below the atomic rate (MI355X_MICROARCH, global float atomics, contention row).
    // Hand-off (cdna_hip_programming.md Guideline 16 R1, counter form): partial rows stored write-through -> every wave's
    // stores drained -> workgroup barrier -> ticket by one lane; the block that draws the last ticket acquires (agent
    // scope: invalidates this CU's L1) and reads the other blocks' rows.  The ticket word is reset by its last
    // user, so the workspace only has to be zero when it is created.
    if (ws) {
        __shared__ int s_last;
        unsigned* cnt = (unsigned*)ws;                       // BPM_MAX_GROUP tickets, then the partial rows
        float* rows = ws + LN_WS_HEAD;
        // partial rows go out WRITE-THROUGH (sc1, 16 bytes per lane): no release fence is needed then -- an agent-scope
        // release (buffer_wbl2) per block would write back the whole L2's dirty dx lines, measured +50 us per launch
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(rows + (size_t)blockIdx.x * 3 * d), 0, 3 * d * 4, 0x00020000);
        for (int v = threadIdx.x; v < 3 * nch; v += NT) {
            const int a = v / nch, c = 4 * (v - a * nch);
            f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) s4 += *(const f32x4*)(&red[a][w][c]);
            __builtin_amdgcn_raw_buffer_store_b128((u32x4&)s4, rw, (a * d + c) * 4, 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int pi = 0;                                          // (not &P - grp.p: taking that address sends the kernel arguments to scratch)
#pragma unroll 1
        for (int i = 1; i < grp.n; ++i)
            if (blockIdx.x >= grp.blk0[i]) pi = i;
        if (threadIdx.x == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(cnt + pi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = ticket == nblk - 1;
            if (s_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(cnt + pi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (!s_last) return;
        // 3 d / 4 four-column slots (array a, columns 4 c4 ..), 16 partial rows in flight per thread
        const float* first = rows + (size_t)grp.blk0[pi] * 3 * d;
        const int nslot = 3 * nch;
        for (int v = threadIdx.x; v < nslot; v += NT) {
            const int a = v / nch, c4 = v - a * nch;
            if ((a < 2 && !want_g) || (a == 2 && !want_c)) continue;
            const float* src = first + a * d + 4 * c4;
            f32x4 acc4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
            for (unsigned b = 0; b < nblk; ++b) acc4 += *(const f32x4*)(src + (size_t)b * 3 * d);
            float* dst = (a == 0 ? P.dgamma : a == 1 ? P.dbeta : P.csum) + 4 * c4;
            *(f32x4*)dst = *(const f32x4*)dst + acc4;
        }
        return;
    }
    for (int c = threadIdx.x; c < d; c += NT) {
        float sg = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { sg += red[0][w][c]; sb += red[1][w][c]; sc += red[2][w][c]; }
        if (want_g) { atomicAdd(P.dgamma + c, sg); atomicAdd(P.dbeta + c, sb); }
        if (want_c) atomicAdd(P.csum + c, sc);
    }
}

// ---------------------------------------------------------------------------
// rows_cast: y = (a [+ b]) * drop_mult ; a is fp32 or CT; written as CT
// (padded) and/or fp32; optional column sums (bias gradient) by atomics.
// Block = 32 rows x 256 columns.
// ---------------------------------------------------------------------------
constexpr int CAST_ROWS = 32;

struct CastP {
    const void* a; int lda; int a_is_ct; const float* b; int ldb;
    void* dct; int ldd; int ctw; float* df32; int ldf; float* colsum;
    int R, C; int cblk;   // column blocks
    DropCfg drop;
};

template <typename CT>
__global__ __launch_bounds__(NT) void rows_cast_kernel(const Grp<CastP> grp) {
    unsigned bid = blockIdx.x, nblk;
    const CastP& P = pick(grp, bid, nblk);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const int c = (bid % P.cblk) * NT + threadIdx.x;
    const int r0 = (bid / P.cblk) * CAST_ROWS;
    const int r1 = min(P.R, r0 + CAST_ROWS);
    const int cmax = P.dct ? P.ctw : P.C;
    if (c >= cmax) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
        float v = 0.f;
        if (c < P.C) {
            v = P.a_is_ct ? Tr<CT>::to_f(((const CT*)P.a)[(size_t)r * P.lda + c]) : ((const float*)P.a)[(size_t)r * P.lda + c];
            if (P.b) v += P.b[(size_t)r * P.ldb + c];
            v *= bpm_drop_mult(drop, (uint32_t)r * (uint32_t)P.C + (uint32_t)c);
            if (P.df32) P.df32[(size_t)r * P.ldf + c] = v;
            s += v;
        }
        if (P.dct) put<CT>(P.dct, (size_t)r * P.ldd + c, v);
    }
    if (P.colsum && c < P.C) atomicAdd(P.colsum + c, s);
}

// Column sums only (bias gradients from a gradient matrix already in HBM): 4-column vector loads, several rows
// per pass and four passes in flight per thread, LDS reduction, one atomic per column and workgroup.
constexpr int CSUM_ROWS = 128;

template <typename CT>
__global__ __launch_bounds__(NT) void colsum_kernel(const Grp<CastP> grp) {
    __shared__ float red[NT * 4];
    unsigned bid = blockIdx.x, nblk;
    const CastP& P = pick(grp, bid, nblk);
    const int CH = (P.C + 3) >> 2;                  // 4-column chunks per row (host: CH <= NT, rows padded to 4*CH)
    const int rpp = NT / CH;                        // rows per pass
    const int rr = threadIdx.x / CH, ch = threadIdx.x - rr * CH;
    const int r0 = bid * CSUM_ROWS, r1 = min(P.R, r0 + CSUM_ROWS);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rr < rpp) {
        auto ld = [&](int r) -> f32x4 {
            const int rc = r < r1 ? r : r0;         // clamped, unconditional load; zeroed below
            f32x4 v;
            if (P.a_is_ct && sizeof(CT) == 2) {
                const bf16x4 t = *(const bf16x4*)((const bf16_t*)P.a + (size_t)rc * P.lda + 4 * ch);
                v = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
            } else {
                v = *(const f32x4*)((const float*)P.a + (size_t)rc * P.lda + 4 * ch);
            }
            return r < r1 ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        for (int r = r0 + rr; r < r1; r += 4 * rpp) {
            const f32x4 a0 = ld(r), a1 = ld(r + rpp), a2 = ld(r + 2 * rpp), a3 = ld(r + 3 * rpp);
            acc += (a0 + a1) + (a2 + a3);
        }
    }
    *(f32x4*)(red + 4 * threadIdx.x) = acc;
    __syncthreads();
    for (int c = threadIdx.x; c < P.C; c += NT) {
        float sum = 0.f;
        for (int q = 0; q < rpp; ++q) sum += red[4 * (q * CH) + c];
        atomicAdd(P.colsum + c, sum);
    }
}

// ---------------------------------------------------------------------------
// Fusion-GMU gating (reference mmtr.py:189-195)
//   out = z*tanh(a1)*x1 + (1-z)*tanh(a2)*x2,  z = sigmoid(ag)
// a1,a2,ag are the three bias-free linear maps produced by the GEMM kernel.
// ---------------------------------------------------------------------------
struct GmuP {
    const float* a1; const float* a2; const float* ag; const float* x1; const float* x2; float* out;
    const float* dout; void* da1; void* da2; void* dag; int ldg; float* dx1; float* dx2; int R;
};

BPM_DEV float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ void gmu2_fwd_kernel(const Grp<GmuP> grp, int d) {
    unsigned bid = blockIdx.x, nblk;
    const GmuP& P = pick(grp, bid, nblk);
    const size_t total = (size_t)P.R * d;
    for (size_t i = (size_t)bid * NT + threadIdx.x; i < total; i += (size_t)nblk * NT) {
        const float z = sigmoidf_(P.ag[i]);
        P.out[i] = z * tanhf(P.a1[i]) * P.x1[i] + (1.f - z) * tanhf(P.a2[i]) * P.x2[i];
    }
}

// da1, da2, dag -> CT [R, ldg] (GEMM operands, pad zeroed); dx1, dx2 (direct terms) -> fp32 [R, d]
template <typename CT>
__global__ void gmu2_bwd_kernel(const Grp<GmuP> grp, int d) {
    unsigned bid = blockIdx.x, nblk;
    const GmuP& P = pick(grp, bid, nblk);
    const size_t total = (size_t)P.R * P.ldg;
    for (size_t i = (size_t)bid * NT + threadIdx.x; i < total; i += (size_t)nblk * NT) {
        const int c = (int)(i % P.ldg);
        const size_t r = i / P.ldg;
        float g1 = 0.f, g2 = 0.f, gz = 0.f;
        if (c < d) {
            const size_t k = r * d + c;
            const float go = P.dout[k];
            const float z = sigmoidf_(P.ag[k]);
            const float h1 = tanhf(P.a1[k]), h2 = tanhf(P.a2[k]);
            const float u1 = P.x1[k], u2 = P.x2[k];
            g1 = go * z * u1 * (1.f - h1 * h1);
            g2 = go * (1.f - z) * u2 * (1.f - h2 * h2);
            gz = go * (h1 * u1 - h2 * u2) * z * (1.f - z);
            P.dx1[k] = go * z * h1;
            P.dx2[k] = go * (1.f - z) * h2;
        }
        put<CT>(P.da1, i, g1);
        put<CT>(P.da2, i, g2);
        put<CT>(P.dag, i, gz);
    }
}

// ---------------------------------------------------------------------------
// out = sum_j in[j] over `count` fp32 elements (count % 4 == 0, 16-byte aligned; out may be one of the inputs).
// The sums of gradient contributions that meet at a tensor several consumers read: a level-1 output feeds a Fusion-GMU
// twice and a level-2 encoder as key and value source; a projected input feeds up to eight encoders.
// ---------------------------------------------------------------------------
struct AddP { float* out; const float* in[BPM_ADDN_MAX]; int n_in; unsigned count4, tail; };

__global__ __launch_bounds__(NT) void add_n_kernel(const Grp<AddP> grp) {
    unsigned bid = blockIdx.x, nblk;
    const AddP& P = pick(grp, bid, nblk);
    for (unsigned q = bid * NT + threadIdx.x; q < P.count4; q += nblk * NT) {
        f32x4 acc = *(const f32x4*)(P.in[0] + 4 * (size_t)q);
#pragma unroll
        for (int j = 1; j < BPM_ADDN_MAX; ++j)
            if (j < P.n_in) acc += *(const f32x4*)(P.in[j] + 4 * (size_t)q);
        *(f32x4*)(P.out + 4 * (size_t)q) = acc;
    }
    if (bid == 0 && threadIdx.x < P.tail) {            // count % 4 trailing elements
        const size_t i = 4 * (size_t)P.count4 + threadIdx.x;
        float a = P.in[0][i];
        for (int j = 1; j < P.n_in; ++j) a += P.in[j][i];
        P.out[i] = a;
    }
}

// ---------------------------------------------------------------------------
// split: fp32 [R, C] (row stride ld) -> bf16 [R, 2 ldp]: columns [0, ldp) hold hi = bf16(x), columns [ldp, 2 ldp) hold
// lo = bf16(x - hi), pad columns [C, ldp) of both planes zero.  Operands of the "bf16x3" products (bpm_gemm_grouped with
// dtype BPM_BF16X3: x y ~ hi hi + hi lo + lo hi on the bf16 MFMA with f32 accumulation: ~2^-16 relative per product).
// ---------------------------------------------------------------------------
struct SplitP { const float* src; bf16_t* dst; int R, C, ld, ldp; };

__global__ __launch_bounds__(NT) void split_rows_kernel(const Grp<SplitP> grp) {
    unsigned bid = blockIdx.x, nblk;
    const SplitP& P = pick(grp, bid, nblk);
    const unsigned cpr = (unsigned)P.ldp >> 2;                     // 4-column chunks per row (ldp % 4 == 0)
    const size_t total = (size_t)P.R * cpr;
    const bool vec = ((P.ld & 3) == 0) && ((((uintptr_t)P.src) & 15) == 0);
    for (size_t i = (size_t)bid * NT + threadIdx.x; i < total; i += (size_t)nblk * NT) {
        const size_t r = i / cpr;
        const int c = 4 * (int)(i - r * cpr);
        f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* sp = P.src + r * P.ld + c;
        if (vec && c + 3 < P.C) x = *(const f32x4*)sp;
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (c + q < P.C) x[q] = sp[q];
        }
        bf16x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) { hi[q] = (bf16_t)x[q]; lo[q] = (bf16_t)(x[q] - (float)hi[q]); }
        bf16_t* dp = P.dst + r * 2 * (size_t)P.ldp + c;
        *(bf16x4*)dp = hi;
        *(bf16x4*)(dp + P.ldp) = lo;
    }
}

constexpr unsigned CAP = 2048;   // blocks per problem for grid-stride kernels

template <typename P> inline bool grp_ok(int n) { return n >= 1 && n <= BPM_MAX_GROUP; }

}  // namespace

#define BPM_DISPATCH_CT(dtype, KERNEL, ...)                                                     \
    do {                                                                                        \
        if ((dtype) == BPM_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, __VA_ARGS__);               \
        else hipLaunchKernelGGL(KERNEL<float>, __VA_ARGS__);                                    \
    } while (0)

extern "C" int bpm_pack_rows_fwd(int dtype, const bpm_pack_problem* q, int n, uint64_t seed, void* stream) {
    if (!q || n < 1 || n > BPM_MAX_GROUP) return BPM_ERR_ARG;
    Grp<PackP> g;
    g.n = n; g.blk0[0] = 0; g.seedp = bpm_seed_ptr(seed);
    for (int i = 0; i < n; ++i) {
        if (!q[i].src || !q[i].dst || q[i].B < 1 || q[i].T < 1 || q[i].C < 1 || q[i].ld < q[i].C) return BPM_ERR_ARG;
        PackP& p = g.p[i];
        p.src = q[i].src; p.dst = q[i].dst; p.dsrc = nullptr; p.g = nullptr; p.ldg = 0;
        p.B = q[i].B; p.T = q[i].T; p.C = q[i].C; p.ld = q[i].ld;
        p.drop = make_drop(q[i].drop_p, seed, q[i].drop_site);
        g.blk0[i + 1] = g.blk0[i] + blocks_for((size_t)p.T * p.B * p.ld, NT * 4, CAP);
    }
    BPM_DISPATCH_CT(dtype, pack_rows_fwd_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_pack_rows_bwd(const bpm_pack_problem* q, int n, uint64_t seed, void* stream) {
    if (!q || n < 1 || n > BPM_MAX_GROUP) return BPM_ERR_ARG;
    Grp<PackP> g;
    g.n = n; g.blk0[0] = 0; g.seedp = bpm_seed_ptr(seed);
    for (int i = 0; i < n; ++i) {
        if (!q[i].g || !q[i].dsrc || q[i].B < 1 || q[i].T < 1 || q[i].C < 1 || q[i].ldg < q[i].C) return BPM_ERR_ARG;
        PackP& p = g.p[i];
        p.src = nullptr; p.dst = nullptr; p.dsrc = q[i].dsrc; p.g = q[i].g; p.ldg = q[i].ldg;
        p.B = q[i].B; p.T = q[i].T; p.C = q[i].C; p.ld = 0;
        p.drop = make_drop(q[i].drop_p, seed, q[i].drop_site);
        g.blk0[i + 1] = g.blk0[i] + blocks_for((size_t)p.T * p.B * p.C, NT * 4, CAP);
    }
    hipLaunchKernelGGL(pack_rows_bwd_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_pack_weights(int dtype, const bpm_pack_desc* table_dev, int ndesc, unsigned total_blocks, void* stream) {
    if (!table_dev || ndesc < 1 || total_blocks < 1) return BPM_ERR_ARG;
    BPM_DISPATCH_CT(dtype, pack_weights_kernel, dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, table_dev, ndesc);
    BPM_CHECK_LAUNCH();
    return 0;
}

static int fill_embed(Grp<EmbP>& g, const bpm_embed_problem* q, int n, int d, uint64_t seed) {
    if (!q || n < 1 || n > BPM_MAX_GROUP || d < 1) return BPM_ERR_ARG;
    g.n = n; g.blk0[0] = 0; g.seedp = bpm_seed_ptr(seed);
    for (int i = 0; i < n; ++i) {
        if (!q[i].x || !q[i].out || q[i].T < 1 || q[i].B < 1) return BPM_ERR_ARG;
        EmbP& p = g.p[i];
        p.x = q[i].x; p.out = q[i].out; p.T = q[i].T; p.B = q[i].B; p.accumulate = q[i].accumulate;
        if (q[i].pos0 < 0 || q[i].pos_stride < 0) return BPM_ERR_ARG;
        p.pos0 = q[i].pos0; p.pstride = q[i].pos_stride > 0 ? q[i].pos_stride : 1;
        p.drop = make_drop(q[i].drop_p, seed, q[i].drop_site);
        g.blk0[i + 1] = g.blk0[i] + blocks_for((size_t)p.T * p.B * d, NT * 4, CAP);
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Fused Adam over one flat fp32 buffer (torch.optim.Adam semantics, no amsgrad):
//   g = grad * grad_scale + wd * p;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),   bc_i = 1 - b_i^t  (passed in)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, size_t n4, float lr_c, float b1, float b2, float eps,
                                                 float wd, float rsq_bc2, float gscale, int zero_grad) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) {
        f32x4 pp = ((f32x4*)p)[i], gg = ((f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
        gg = gg * gscale + wd * pp;
        mm = b1 * mm + (1.f - b1) * gg;
        vv = b2 * vv + (1.f - b2) * gg * gg;
#pragma unroll
        for (int q = 0; q < 4; ++q) pp[q] -= lr_c * mm[q] / (sqrtf(vv[q]) * rsq_bc2 + eps);
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
        if (zero_grad) ((f32x4*)g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

extern "C" int bpm_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n == 0 || (n & 3) || step < 1) return BPM_ERR_ARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return BPM_ERR_ALIGN;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const size_t n4 = n / 4;
    const unsigned blocks = blocks_for(n4, NT, 256 * 16);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n4,
                       (float)(lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale, zero_grad);
    BPM_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// The same step, table driven, writing the CT weight shadows as it stores the updated masters (no second pass over the
// flat master for the shadow refresh: 2.7 GB read + 1.35 GB written per optimizer step at hidden 768).  The flat buffer
// is cut into segments (device-resident table, built once): runs of parameters without a plain shadow, and one segment per
// parameter that has one -- a whole [rows, cols] matrix whose shadow is [rows, dst_ld], pad columns left as they are
// (zero since allocation).  A block covers ADAM_CHUNK consecutive f32x4 of ONE segment, so the segment is looked up once
// per block; every thread has its 4 x 4 loads in flight before the first use.
// ---------------------------------------------------------------------------
constexpr int ADAM_ITER = 4;
constexpr int ADAM_CHUNK = NT * ADAM_ITER;             // f32x4 per block

template <typename CT>
__global__ __launch_bounds__(NT) void adam_table_kernel(const bpm_adam_seg* __restrict__ tab, int nseg, float* __restrict__ p,
                                                       float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                       float lr_c, float b1, float b2, float eps, float wd, float rsq_bc2,
                                                       float gscale, int zero_grad) {
    const bpm_adam_seg S = find_desc(tab, nseg, blockIdx.x);
    const size_t base = S.off4 + (size_t)(blockIdx.x - S.blk0) * ADAM_CHUNK;
    const size_t end = S.off4 + S.n4;
    f32x4 pp[ADAM_ITER], gg[ADAM_ITER], mm[ADAM_ITER], vv[ADAM_ITER];
#pragma unroll
    for (int j = 0; j < ADAM_ITER; ++j) {
        const size_t i = base + j * NT + threadIdx.x;
        const size_t ic = i < end ? i : S.off4;             // clamped, unconditional loads
        pp[j] = ((const f32x4*)p)[ic]; gg[j] = ((const f32x4*)g)[ic]; mm[j] = ((const f32x4*)m)[ic]; vv[j] = ((const f32x4*)v)[ic];
    }
#pragma unroll
    for (int j = 0; j < ADAM_ITER; ++j) {
        const size_t i = base + j * NT + threadIdx.x;
        if (i >= end) continue;
        f32x4 x = pp[j], gr = gg[j] * gscale + wd * pp[j];
        const f32x4 mo = b1 * mm[j] + (1.f - b1) * gr;
        const f32x4 vo = b2 * vv[j] + (1.f - b2) * gr * gr;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] -= lr_c * mo[q] / (sqrtf(vo[q]) * rsq_bc2 + eps);
        ((f32x4*)p)[i] = x; ((f32x4*)m)[i] = mo; ((f32x4*)v)[i] = vo;
        if (zero_grad) ((f32x4*)g)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (S.dst) {
            const size_t e = 4 * (i - S.off4);               // element index inside the [rows, cols] matrix; cols % 4 == 0
            if (e < (size_t)S.rows * S.cols) {               // (the segment's 64-element alignment tail has no shadow)
                size_t o = e;
                if (S.cols != S.dst_ld) { const size_t r = e / (unsigned)S.cols; o = r * S.dst_ld + (e - r * S.cols); }
                put4<CT>(S.dst, o, x);
            }
        }
    }
}

extern "C" int bpm_adam_blocks(size_t n4) { return (int)((n4 + ADAM_CHUNK - 1) / ADAM_CHUNK); }

extern "C" int bpm_adam_step_table(int dtype, const bpm_adam_seg* table_dev, int nseg, unsigned total_blocks, float* param, float* grad,
                                   float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, int step, float grad_scale, int zero_grad, void* stream) {
    if (!table_dev || nseg < 1 || total_blocks < 1 || !param || !grad || !exp_avg || !exp_avg_sq || step < 1) return BPM_ERR_ARG;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return BPM_ERR_ALIGN;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    BPM_DISPATCH_CT(dtype, adam_table_kernel, dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, table_dev, nseg, param, grad,
                    exp_avg, exp_avg_sq, (float)(lr / bc1), beta1, beta2, eps, weight_decay, (float)(1.0 / sqrt(bc2)), grad_scale,
                    zero_grad);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_fold_bias(const bpm_fold_desc* table_dev, int ndesc, unsigned total_blocks, void* stream) {
    if (!table_dev || ndesc < 1 || total_blocks < 1) return BPM_ERR_ARG;
    hipLaunchKernelGGL(fold_bias_kernel, dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, table_dev, ndesc);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_unfold_grads(const bpm_unfold_desc* table_dev, int ndesc, unsigned total_blocks, int store_dw, void* stream) {
    if (!table_dev || ndesc < 1 || total_blocks < 1) return BPM_ERR_ARG;
    hipLaunchKernelGGL(unfold_grads_kernel, dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, table_dev, ndesc, store_dw);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_zero_segment_blocks(unsigned n) { return (int)((n + ZSEG - 1) / ZSEG); }

extern "C" int bpm_zero_segments(const bpm_zero_desc* table_dev, int ndesc, unsigned total_blocks, void* stream) {
    if (!table_dev || ndesc < 1 || total_blocks < 1) return BPM_ERR_ARG;
    hipLaunchKernelGGL(zero_segments_kernel, dim3(total_blocks), dim3(NT), 0, (hipStream_t)stream, table_dev, ndesc);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_embed_pos_fwd(const bpm_embed_problem* q, int n, const float* table, int table_rows, int d,
                                 float scale, uint64_t seed, void* stream) {
    Grp<EmbP> g;
    int rc = fill_embed(g, q, n, d, seed);
    if (rc) return rc;
    if (!table) return BPM_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (table_rows < q[i].pos0 + (q[i].T - 1) * (q[i].pos_stride > 0 ? q[i].pos_stride : 1) + 2) return BPM_ERR_ARG;
    hipLaunchKernelGGL(embed_pos_fwd_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g, table, d, scale);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_embed_pos_bwd(const bpm_embed_problem* q, int n, int d, float scale, uint64_t seed, void* stream) {
    Grp<EmbP> g;
    int rc = fill_embed(g, q, n, d, seed);
    if (rc) return rc;
    hipLaunchKernelGGL(embed_pos_bwd_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g, d, scale);
    BPM_CHECK_LAUNCH();
    return 0;
}

static int fill_ln(Grp<LnP>& g, const bpm_ln_problem* q, int n, int d, bool bwd, int* span, uint64_t seed = 0) {
    if (!q || n < 1 || n > BPM_MAX_GROUP || d < 1 || d > 64 * MAXE) return BPM_ERR_ARG;
    g.n = n; g.blk0[0] = 0; g.seedp = bpm_seed_ptr(seed);
    *span = d;
    for (int i = 0; i < n; ++i) {
        const bpm_ln_problem& s = q[i];
        LnP& p = g.p[i];
        if (!s.x || !s.gamma || !s.mean || !s.rstd || s.R < 1) return BPM_ERR_ARG;
        if (bwd) {
            if (!s.dy || !s.dx || s.ldy < d || ((s.dgamma == nullptr) != (s.dbeta == nullptr))) return BPM_ERR_ARG;
            if (s.cast && (s.ldc < d || s.ldc > 64 * ((d + 63) / 64))) return BPM_ERR_ARG;
            if (!s.cast && s.cast_colsum) return BPM_ERR_ARG;
        } else {
            if (!s.beta || !s.out || s.ldo < d || s.ldo > 64 * MAXE) return BPM_ERR_ARG;
            if (!s.out_f32 && s.ldo > *span) *span = s.ldo;
        }
        p.x = s.x; p.gamma = s.gamma; p.beta = s.beta; p.out = s.out; p.ldo = s.ldo; p.out_f32 = s.out_f32;
        p.mean = s.mean; p.rstd = s.rstd; p.R = s.R;
        p.dy = s.dy; p.ldy = s.ldy; p.add = s.add; p.dx = s.dx; p.dgamma = s.dgamma; p.dbeta = s.dbeta;
        p.cast = bwd ? s.cast : nullptr; p.ldc = s.ldc; p.csum = s.cast_colsum;
        p.drop = make_drop(bwd && s.cast ? s.drop_p : 0.f, seed, s.drop_site);
        g.blk0[i + 1] = g.blk0[i] + blocks_for((size_t)s.R, 4, bwd ? 256 : 4096);
    }
    return 0;
}

// vector kernels: d % 4 == 0, every row 16-byte (CT outputs: 8-byte) aligned
static bool ln_vec_ok(const bpm_ln_problem* q, int n, int d, bool bwd, int ct_size) {
    if (d % 4 || d > 4 * 64 * 4) return false;
    for (int i = 0; i < n; ++i) {
        const bpm_ln_problem& s = q[i];
        uintptr_t a = (uintptr_t)s.x | (uintptr_t)s.gamma;
        if (bwd) {
            a |= (uintptr_t)s.dy | (uintptr_t)s.add | (uintptr_t)s.dx;
            if (s.ldy % 4) return false;
            if (s.cast && (((uintptr_t)s.cast % (4 * ct_size)) || s.ldc % 4)) return false;
        } else {
            a |= (uintptr_t)s.beta;
            if (s.ldo % 4 || ((uintptr_t)s.out % (4 * (s.out_f32 ? 4 : ct_size)))) return false;
        }
        if (a & 15) return false;
    }
    return true;
}

extern "C" int bpm_ln_fwd(int dtype, const bpm_ln_problem* q, int n, int d, float eps, void* stream) {
    Grp<LnP> g;
    int span;
    int rc = fill_ln(g, q, n, d, false, &span);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (ln_vec_ok(q, n, d, false, dtype == BPM_BF16 ? 2 : 4) && span <= 4 * 64 * 4) {
#define BPM_LN_FWDV(NV)                                                                                        \
        if (span <= 256 * NV) {                                                                                \
            if (dtype == BPM_BF16) hipLaunchKernelGGL((ln_fwd_vec_kernel<bf16_t, NV>), dim3(g.blk0[n]), dim3(NT), 0, s, g, d, eps); \
            else hipLaunchKernelGGL((ln_fwd_vec_kernel<float, NV>), dim3(g.blk0[n]), dim3(NT), 0, s, g, d, eps);                   \
            BPM_CHECK_LAUNCH();                                                                                \
            return 0;                                                                                          \
        }
        BPM_LN_FWDV(1) BPM_LN_FWDV(2) BPM_LN_FWDV(3) BPM_LN_FWDV(4)
#undef BPM_LN_FWDV
    }
#define BPM_LN_FWD(NE)                                                                            \
    if (span <= 64 * NE) {                                                                        \
        if (dtype == BPM_BF16) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, NE>), dim3(g.blk0[n]), dim3(NT), 0, s, g, d, eps); \
        else hipLaunchKernelGGL((ln_fwd_kernel<float, NE>), dim3(g.blk0[n]), dim3(NT), 0, s, g, d, eps);                    \
        BPM_CHECK_LAUNCH();                                                                       \
        return 0;                                                                                 \
    }
    BPM_LN_FWD(1) BPM_LN_FWD(2) BPM_LN_FWD(5) BPM_LN_FWD(8) BPM_LN_FWD(12) BPM_LN_FWD(16) BPM_LN_FWD(24) BPM_LN_FWD(32)
#undef BPM_LN_FWD
    return BPM_ERR_ARG;
}

extern "C" size_t bpm_ln_bwd_ws_bytes(int n, int d) {
    return ((size_t)(n > 0 ? n : 0) * 128 * 3 * (size_t)(d > 0 ? d : 0) + LN_WS_HEAD) * sizeof(float);
}

extern "C" int bpm_ln_bwd_ws(int dtype, const bpm_ln_problem* q, int n, int d, uint64_t seed, void* ws, size_t ws_bytes, void* stream) {
    if (dtype != BPM_F32 && dtype != BPM_BF16) return BPM_ERR_ARG;
    Grp<LnP> g;
    int span;
    int rc = fill_ln(g, q, n, d, true, &span, seed);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (ln_vec_ok(q, n, d, true, dtype == BPM_BF16 ? 2 : 4)) {
        // two rows per wave and iteration: half the blocks of the scalar kernel
        // at most 64 blocks per problem (= partial rows the last block sums): 128 took 78 / 98 us stand-alone at hidden 768,
        // six problems (with parameter gradients / with the fused cast), 64 takes 72 / 88
        for (int i = 0; i < n; ++i) g.blk0[i + 1] = g.blk0[i] + blocks_for((size_t)q[i].R, 8, 64);
        int maxw = d;
        bool sums = false;
        for (int i = 0; i < n; ++i) {
            if (q[i].cast && q[i].ldc > maxw) maxw = q[i].ldc;
            sums = sums || q[i].dgamma || (q[i].cast && q[i].cast_colsum);
            if (((uintptr_t)q[i].dgamma | (uintptr_t)q[i].dbeta | (uintptr_t)q[i].cast_colsum) & 15) ws = nullptr;   // 16-byte row sums
        }
        // the last block of a problem adds its partial rows to dgamma / dbeta / cast_colsum with a plain read-modify-write:
        // exact only while no other problem of this launch owns the same row (and, host contract, no other stream
        // accumulates into it while this launch runs: include/bpmult_hip.h).  Shared rows go out as float atomics.
        if (ws && sums) {
            const float* rows[3 * BPM_MAX_GROUP];
            int nr = 0;
            for (int i = 0; i < n; ++i) {
                if (q[i].dgamma) { rows[nr++] = q[i].dgamma; rows[nr++] = q[i].dbeta; }
                if (q[i].cast && q[i].cast_colsum) rows[nr++] = q[i].cast_colsum;
            }
            for (int a = 0; a < nr && ws; ++a)
                for (int b = a + 1; b < nr; ++b)
                    if (rows[a] && rows[a] == rows[b]) { ws = nullptr; break; }
        }
        float* w = nullptr;
        if (ws && sums) {
            if (((uintptr_t)ws & 15) || ws_bytes < ((size_t)g.blk0[n] * 3 * d + LN_WS_HEAD) * sizeof(float)) return BPM_ERR_ARG;
            w = (float*)ws;
        }
#define BPM_LN_BWDV(NV)                                                                                        \
        if (maxw <= 256 * NV) {                                                                                \
            if (dtype == BPM_BF16) hipLaunchKernelGGL((ln_bwd_vec_kernel<bf16_t, NV>), dim3(g.blk0[n]), dim3(NT), 0, s, g, d, w); \
            else hipLaunchKernelGGL((ln_bwd_vec_kernel<float, NV>), dim3(g.blk0[n]), dim3(NT), 0, s, g, d, w);                   \
            BPM_CHECK_LAUNCH();                                                                                \
            return 0;                                                                                          \
        }
        BPM_LN_BWDV(1) BPM_LN_BWDV(2) BPM_LN_BWDV(3) BPM_LN_BWDV(4)
#undef BPM_LN_BWDV
    }
#define BPM_LN_BWD(NE)                                                                                               \
    if (d <= 64 * NE) {                                                                                              \
        if (dtype == BPM_BF16) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, NE>), dim3(g.blk0[n]), dim3(NT), 0, s, g, d); \
        else hipLaunchKernelGGL((ln_bwd_kernel<float, NE>), dim3(g.blk0[n]), dim3(NT), 0, s, g, d);                   \
        BPM_CHECK_LAUNCH();                                                                                          \
        return 0;                                                                                                    \
    }
    BPM_LN_BWD(1) BPM_LN_BWD(2) BPM_LN_BWD(5) BPM_LN_BWD(8) BPM_LN_BWD(12) BPM_LN_BWD(16) BPM_LN_BWD(24) BPM_LN_BWD(32)
#undef BPM_LN_BWD
    return BPM_ERR_ARG;
}

extern "C" int bpm_ln_bwd(int dtype, const bpm_ln_problem* q, int n, int d, uint64_t seed, void* stream) {
    return bpm_ln_bwd_ws(dtype, q, n, d, seed, nullptr, 0, stream);
}

extern "C" int bpm_rows_cast(int dtype, const bpm_cast_problem* q, int n, uint64_t seed, void* stream) {
    if (!q || n < 1 || n > BPM_MAX_GROUP) return BPM_ERR_ARG;
    Grp<CastP> g;
    g.n = n; g.blk0[0] = 0; g.seedp = bpm_seed_ptr(seed);
    for (int i = 0; i < n; ++i) {
        const bpm_cast_problem& s = q[i];
        if (!s.a || s.R < 1 || s.C < 1 || s.lda < s.C || (s.b && s.ldb < s.C) || (s.dst_ct && s.ldd < s.C) ||
            (s.dst_f32 && s.ldf < s.C) || (!s.dst_ct && !s.dst_f32 && !s.colsum))
            return BPM_ERR_ARG;
        CastP& p = g.p[i];
        p.a = s.a; p.lda = s.lda; p.a_is_ct = s.a_is_ct; p.b = s.b; p.ldb = s.ldb;
        p.dct = s.dst_ct; p.ldd = s.ldd; p.ctw = s.ct_cols > 0 ? s.ct_cols : s.ldd; p.df32 = s.dst_f32; p.ldf = s.ldf; p.colsum = s.colsum;
        p.R = s.R; p.C = s.C;
        if (s.dst_ct && (p.ctw < s.C || p.ctw > s.ldd)) return BPM_ERR_ARG;
        const int cols = s.dst_ct ? p.ctw : s.C;
        p.cblk = (cols + NT - 1) / NT;
        p.drop = make_drop(s.drop_p, seed, s.drop_site);
        g.blk0[i + 1] = g.blk0[i] + (unsigned)p.cblk * ((s.R + CAST_ROWS - 1) / CAST_ROWS);
    }
    // all problems "column sums of an aligned matrix": the vectorised reduction kernel
    bool only_sums = true;
    for (int i = 0; i < n; ++i) {
        const bpm_cast_problem& s = q[i];
        const int ch = (s.C + 3) / 4;
        const size_t esz = (s.a_is_ct && dtype == BPM_BF16) ? 2 : 4;
        only_sums = only_sums && !s.dst_ct && !s.dst_f32 && !s.b && s.colsum && !(s.drop_p > 0.f) && ch <= NT &&
                    s.lda >= 4 * ch && (s.lda % 4) == 0 && ((uintptr_t)s.a % (4 * esz)) == 0;
    }
    if (only_sums) {
        for (int i = 0; i < n; ++i) g.blk0[i + 1] = g.blk0[i] + (unsigned)((q[i].R + CSUM_ROWS - 1) / CSUM_ROWS);
        BPM_DISPATCH_CT(dtype, colsum_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g);
        BPM_CHECK_LAUNCH();
        return 0;
    }
    BPM_DISPATCH_CT(dtype, rows_cast_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g);
    BPM_CHECK_LAUNCH();
    return 0;
}

static int fill_gmu(Grp<GmuP>& g, const bpm_gmu_problem* q, int n, int d, bool bwd) {
    if (!q || n < 1 || n > BPM_MAX_GROUP || d < 1) return BPM_ERR_ARG;
    g.n = n; g.blk0[0] = 0; g.seedp = nullptr;
    for (int i = 0; i < n; ++i) {
        const bpm_gmu_problem& s = q[i];
        if (!s.a1 || !s.a2 || !s.ag || !s.x1 || !s.x2 || s.R < 1) return BPM_ERR_ARG;
        if (bwd ? (!s.dout || !s.da1 || !s.da2 || !s.dag || !s.dx1 || !s.dx2 || s.ldg < d) : !s.out) return BPM_ERR_ARG;
        GmuP& p = g.p[i];
        p.a1 = s.a1; p.a2 = s.a2; p.ag = s.ag; p.x1 = s.x1; p.x2 = s.x2; p.out = s.out;
        p.dout = s.dout; p.da1 = s.da1; p.da2 = s.da2; p.dag = s.dag; p.ldg = s.ldg; p.dx1 = s.dx1; p.dx2 = s.dx2; p.R = s.R;
        g.blk0[i + 1] = g.blk0[i] + blocks_for((size_t)s.R * (bwd ? s.ldg : d), NT * 4, CAP);
    }
    return 0;
}

extern "C" int bpm_gmu2_fwd(const bpm_gmu_problem* q, int n, int d, void* stream) {
    Grp<GmuP> g;
    int rc = fill_gmu(g, q, n, d, false);
    if (rc) return rc;
    hipLaunchKernelGGL(gmu2_fwd_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g, d);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_gmu2_bwd(int dtype, const bpm_gmu_problem* q, int n, int d, void* stream) {
    Grp<GmuP> g;
    int rc = fill_gmu(g, q, n, d, true);
    if (rc) return rc;
    BPM_DISPATCH_CT(dtype, gmu2_bwd_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g, d);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_split_rows(const bpm_split_problem* q, int n, void* stream) {
    if (!q || n < 1 || n > BPM_MAX_GROUP) return BPM_ERR_ARG;
    Grp<SplitP> g;
    g.n = n; g.blk0[0] = 0; g.seedp = nullptr;
    for (int i = 0; i < n; ++i) {
        const bpm_split_problem& s = q[i];
        if (!s.src || !s.dst || s.R < 1 || s.C < 1 || s.ld < s.C || s.ldp < s.C || (s.ldp & 3)) return BPM_ERR_ARG;
        if (((uintptr_t)s.dst & 7) || ((uintptr_t)s.src & 3)) return BPM_ERR_ALIGN;
        SplitP& p = g.p[i];
        p.src = s.src; p.dst = (bf16_t*)s.dst; p.R = s.R; p.C = s.C; p.ld = s.ld; p.ldp = s.ldp;
        g.blk0[i + 1] = g.blk0[i] + blocks_for((size_t)s.R * (s.ldp >> 2), NT * 4, CAP);
    }
    hipLaunchKernelGGL(split_rows_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g);
    BPM_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// expand_heads: head-major q / dO [B, H, T, dhp] -> block rows [(h*T + t)*B + b, ld] whose columns [h dh, (h+1) dh) hold
// the head's vector and all others zero.  With these the per-head products of the low-rank key side (a handful of query
// rows: engine.EncoderGroupPlan) are plain grouped GEMMs over all heads at once: Qexp W_k' = every head's query through
// its block of W_k', Qexp^T U = every head's block of the weight gradient.  Also the value-projection bias gradient,
// sum_{b,t} rowsum(Pd[b,h,t,:]) dO[b,h,t,:] (the column sums of dV = Pd^T dO; rowsum(Pd) != 1 under attention dropout).
// One workgroup per (problem, head); its four waves take the head's T*B rows in turn, the bias sums are combined in a
// fixed order (no atomics: bitwise reproducible) and WRITTEN (the launch owns dbias).
// ---------------------------------------------------------------------------
struct ExpP { const void* q; const void* dO; void* qexp; void* dOexp; const void* Pd; float* dbias; int B, H, T, S, dh, dhp, ld; };

constexpr int EXP_NT = 1024;      // 16 waves, one row each per pass (T*B = 16 rows at the headline: one pass)

template <typename CT>
__global__ __launch_bounds__(EXP_NT) void expand_heads_kernel(const Grp<ExpP> grp) {
    __shared__ float red[EXP_NT / 64][128];
    unsigned bid = blockIdx.x, nblk;
    const ExpP& P = pick(grp, bid, nblk);
    const int h = (int)bid, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = h * P.dh;
    float acc0 = 0.f, acc1 = 0.f;                         // bias sums of columns lane, lane + 64 of this head
    for (int tb = wv; tb < P.T * P.B; tb += EXP_NT / 64) {
        const int t = tb / P.B, b = tb - t * P.B;
        const size_t r = (size_t)(h * P.T + t) * P.B + b;
        const CT* qrow = (const CT*)P.q + ((size_t)(b * P.H + h) * P.T + t) * P.dhp;
        const CT* drow = (const CT*)P.dO + ((size_t)(b * P.H + h) * P.T + t) * P.dhp;
        for (int c = lane; c < P.ld; c += 64) {
            const bool in = c >= c0 && c < c0 + P.dh;
            put<CT>(P.qexp, r * P.ld + c, in ? Tr<CT>::to_f(qrow[c - c0]) : 0.f);
            put<CT>(P.dOexp, r * P.ld + c, in ? Tr<CT>::to_f(drow[c - c0]) : 0.f);
        }
        if (P.dbias) {
            float rs = 0.f;
            const CT* prow = (const CT*)P.Pd + r * P.S;
            for (int j = lane; j < P.S; j += 64) rs += Tr<CT>::to_f(prow[j]);
            rs = wave_sum(rs);
            if (lane < P.dh) acc0 += rs * Tr<CT>::to_f(drow[lane]);
            if (lane + 64 < P.dh) acc1 += rs * Tr<CT>::to_f(drow[lane + 64]);
        }
    }
    if (!P.dbias) return;                                 // uniform per block
    red[wv][lane] = acc0; red[wv][lane + 64] = acc1;
    __syncthreads();
    for (int j = threadIdx.x; j < P.dh; j += EXP_NT) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < EXP_NT / 64; ++w) s += red[w][j];
        P.dbias[c0 + j] = s;
    }
}

extern "C" int bpm_expand_heads(int dtype, const bpm_expand_problem* q, int n, void* stream) {
    if (!q || n < 1 || n > BPM_MAX_GROUP || (dtype != BPM_F32 && dtype != BPM_BF16)) return BPM_ERR_ARG;
    Grp<ExpP> g;
    g.n = n; g.blk0[0] = 0; g.seedp = nullptr;
    for (int i = 0; i < n; ++i) {
        const bpm_expand_problem& s = q[i];
        if (!s.q || !s.dO || !s.qexp || !s.dOexp || s.B < 1 || s.H < 1 || s.T < 1 || s.dh < 1 || s.dh > s.dhp || s.dh > 128 ||
            s.ld < s.H * s.dh || (s.dbias && (!s.Pd || s.S < 1)))
            return BPM_ERR_ARG;
        ExpP& p = g.p[i];
        p.q = s.q; p.dO = s.dO; p.qexp = s.qexp; p.dOexp = s.dOexp; p.Pd = s.Pd; p.dbias = s.dbias;
        p.B = s.B; p.H = s.H; p.T = s.T; p.S = s.S; p.dh = s.dh; p.dhp = s.dhp; p.ld = s.ld;
        g.blk0[i + 1] = g.blk0[i] + (unsigned)s.H;
    }
    if (dtype == BPM_BF16) hipLaunchKernelGGL(expand_heads_kernel<bf16_t>, dim3(g.blk0[n]), dim3(EXP_NT), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(expand_heads_kernel<float>, dim3(g.blk0[n]), dim3(EXP_NT), 0, (hipStream_t)stream, g);
    BPM_CHECK_LAUNCH();
    return 0;
}

extern "C" int bpm_add_n(const bpm_addn_problem* q, int n, void* stream) {
    if (!q || n < 1 || n > BPM_MAX_GROUP) return BPM_ERR_ARG;
    Grp<AddP> g;
    g.n = n; g.blk0[0] = 0; g.seedp = nullptr;
    for (int i = 0; i < n; ++i) {
        const bpm_addn_problem& s = q[i];
        if (!s.out || s.n_in < 1 || s.n_in > BPM_ADDN_MAX || s.count < 1 || s.count > (size_t)1 << 33) return BPM_ERR_ARG;
        AddP& p = g.p[i];
        uintptr_t al = (uintptr_t)s.out;
        for (int j = 0; j < BPM_ADDN_MAX; ++j) {
            p.in[j] = j < s.n_in ? s.src[j] : s.src[0];
            if (j < s.n_in && !s.src[j]) return BPM_ERR_ARG;
            al |= (uintptr_t)p.in[j];
        }
        if (al & 15) return BPM_ERR_ALIGN;
        p.out = s.out; p.n_in = s.n_in; p.count4 = (unsigned)(s.count >> 2); p.tail = (unsigned)(s.count & 3);
        g.blk0[i + 1] = g.blk0[i] + blocks_for((size_t)p.count4, NT * 4, CAP);
    }
    hipLaunchKernelGGL(add_n_kernel, dim3(g.blk0[n]), dim3(NT), 0, (hipStream_t)stream, g);
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
