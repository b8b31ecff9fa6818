// The [B,d]-sized tail of BPMulT (gfx950): level 1 -> 3 residual + first / last token pick (mmtr.py:806-808), the final
// n-way gated fusion TextShifting{3,4}Layer (mmtr.py:197-247, call sites 574 / 857) and the residual MLP head
// proj1 / ReLU / dropout / proj2 / + h / out_layer (mmtr.py:577-583, 860-866), forward and backward.
//
// Shapes: B <= a few hundred rows of d = hidden floats, 33 MB of fp32 weights at hidden 768 (3-modal).  This is
// weight streaming with a handful of rows, not a GEMM: every product runs on the f32 VALU straight from the fp32 master
// weights (exact f32, no shadows), a wave per output column (forward) / a lane per output column (data gradients) /
// a lane per 4 weight-gradient elements, the few activation rows served by L1 / L2.  The phases depend on each
// other through whole [B,d] vectors, so they are separate launches inside ONE entry point per direction (a kernel
// boundary costs ~1.5 us, a grid barrier more: MI355X_MICROARCH price list): 6 launches forward, ~n + 11 backward,
// against ~100 torch / hipBLASLt launches for the same arithmetic.
#include "bpm_common.h"
#include "../../include/bpmult_hip.h"

namespace {

constexpr int TNT = 256;
constexpr int BC = 8;                      // batch rows per register chunk

enum { ACT_NONE = 0, ACT_SIGMOID = 1, ACT_TANH = 2, ACT_RELU_DROP = 3 };

struct LinP {                              // out[b, j] = act(sum_k W[j, k] in[b, k] + bias[j]) (+ resid[b, j])
    const float* W; int ldw;
    const float* in; int ldin;
    const float* bias;
    const float* resid; int ldr;
    float* out; int ldo;
    int N, K, act;
    DropCfg drop;
    int col0;                              // first global column index (blocks are dealt over all problems)
};
struct LinGrp { const uint64_t* seedp; int n, B, total_cols; LinP p[8]; };

// forward skinny product: one wave per output column, lanes over k
__global__ __launch_bounds__(TNT) void tail_linear_kernel(const LinGrp g) {
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * (TNT / 64) + (threadIdx.x >> 6);
    if (col >= g.total_cols) return;
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if (col >= g.p[i].col0) pi = i;
    const LinP& P = g.p[pi];
    const DropCfg drop = bpm_resolve_drop(P.drop, g.seedp);
    const int j = col - P.col0;
    const float* w = P.W + (size_t)j * P.ldw;
    for (int b0 = 0; b0 < g.B; b0 += BC) {
        float acc[BC];
#pragma unroll
        for (int r = 0; r < BC; ++r) acc[r] = 0.f;
        // (four k per lane and three such steps in flight where rows are 16-byte aligned: the one-k loop made a K = 2304
        // column 36 dependent rounds of loads -- 54 us for the first linear layer of the head at hidden 768)
        const bool wide4 = ((P.K | P.ldw | P.ldin) & 3) == 0 && ((((uintptr_t)P.W | (uintptr_t)P.in) & 15) == 0);
        if (wide4) {
#pragma unroll 3
            for (int k = 4 * lane; k < P.K; k += 256) {
                const f32x4 wv = *(const f32x4*)(w + k);
#pragma unroll
                for (int r = 0; r < BC; ++r) {
                    const int b = min(b0 + r, g.B - 1);
                    const f32x4 xv = *(const f32x4*)(P.in + (size_t)b * P.ldin + k);
                    acc[r] += (wv[0] * xv[0] + wv[1] * xv[1]) + (wv[2] * xv[2] + wv[3] * xv[3]);
                }
            }
        } else {
            for (int k = lane; k < P.K; k += 64) {
                const float wv = w[k];
#pragma unroll
                for (int r = 0; r < BC; ++r) {
                    const int b = min(b0 + r, g.B - 1);
                    acc[r] = fmaf(wv, P.in[(size_t)b * P.ldin + k], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < BC; ++r) acc[r] = wave_sum(acc[r]);
        if (lane < BC && b0 + lane < g.B) {
            const int b = b0 + lane;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < BC; ++r) v = lane == r ? acc[r] : v;
            if (P.bias) v += P.bias[j];
            if (P.act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
            else if (P.act == ACT_TANH) v = tanhf(v);
            else if (P.act == ACT_RELU_DROP) v = fmaxf(v, 0.f) * bpm_drop_mult(drop, (uint32_t)b * (uint32_t)P.N + (uint32_t)j);
            if (P.resid) v += P.resid[(size_t)b * P.ldr + j];
            P.out[(size_t)b * P.ldo + j] = v;
        }
    }
}

// x[b, i d + c] = top_i[0,b,c] + mid_i[0,b,c] + top_i[N_i-1,b,c] + mid_i[N_i-1,b,c]   (i < 3);  x[b, 3 d + c] = extra[b,c]
struct PickP { const float* top[3]; const float* mid[3]; const float* extra; float* x; int N[3]; int B, d, n; };
__global__ void tail_pick_kernel(const PickP P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nd = P.n * P.d;
    if (idx >= P.B * nd) return;
    const int b = idx / nd, q = idx % nd, i = q / P.d, c = q % P.d;
    float v;
    if (i < 3) {
        const size_t r0 = (size_t)b * P.d + c, r1 = ((size_t)(P.N[i] - 1) * P.B + b) * P.d + c;
        v = (P.top[i][r0] + P.mid[i][r0]) + (P.top[i][r1] + P.mid[i][r1]);
    } else {
        v = P.extra[(size_t)b * P.d + c];
    }
    P.x[idx] = v;
}

// h[b,c] = sum_i z[b, i d + c] t[b, i d + c]
__global__ void tail_combine_kernel(const float* __restrict__ z, const float* __restrict__ t, float* __restrict__ h, int B, int d, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * d) return;
    const int b = idx / d, c = idx % d;
    float v = 0.f;
    for (int i = 0; i < n; ++i) v = fmaf(z[(size_t)b * n * d + i * d + c], t[(size_t)b * n * d + i * d + c], v);
    h[idx] = v;
}

// data gradient: out[b, k] (+)= sum_j dy[b, j] W[j, k]  (* gate)  (+ add[b, k]);  block = 64 columns x 16 j-groups
struct NnP {
    const float* dy; int lddy; const float* W; int ldw; int J, K;
    float* out; int ldo; int accumulate;
    const float* gate; int ldg; float gate_scale;      // out *= gate[b,k] > 0 ? gate_scale : 0
    const float* add; int lda;
    int blk0;
};
struct NnGrp { int n, B, total_blk; NnP p[8]; };
constexpr int NN_JG = 16;                  // j groups per block (1024 threads = 64 columns x 16 groups)
__global__ __launch_bounds__(64 * NN_JG) void tail_nn_kernel(const NnGrp g) {
    __shared__ float part[NN_JG - 1][BC][64];
    int bid = blockIdx.x, pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if (bid >= g.p[i].blk0) pi = i;
    const NnP& P = g.p[pi];
    bid -= P.blk0;
    const int col = threadIdx.x & 63;
    const int jg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: the dy reads below become scalar loads
    const int k = bid * 64 + col;
    const bool kok = k < P.K;
    for (int b0 = 0; b0 < g.B; b0 += BC) {
        float acc[BC];
#pragma unroll
        for (int r = 0; r < BC; ++r) acc[r] = 0.f;
        if (kok) {
            // eight weight rows and their BC x 8 dy values are requested before the first is used (a rolled loop -- and an
            // `unroll 8` of it -- waited for each row's nine loads in turn: 48 dependent rounds per J = 768 product): 21 -> 18 us;
            // what is left is 12 workgroups with 32 KB in flight each for 2.4 MB of weights -- more workgroups would need a
            // reduction across them
            constexpr int NU = 8;
            const float* dyr[BC];
#pragma unroll
            for (int r = 0; r < BC; ++r) dyr[r] = P.dy + (size_t)min(b0 + r, g.B - 1) * P.lddy;
            for (int j0 = jg; j0 < P.J; j0 += NN_JG * NU) {
                float wv[NU], dv[NU][BC];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int j = min(j0 + u * NN_JG, P.J - 1);           // clamped: surplus steps are masked to zero below
                    wv[u] = P.W[(size_t)j * P.ldw + k];
#pragma unroll
                    for (int r = 0; r < BC; ++r) dv[u][r] = dyr[r][j];
                }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const float w = j0 + u * NN_JG < P.J ? wv[u] : 0.f;
#pragma unroll
                    for (int r = 0; r < BC; ++r) acc[r] = fmaf(dv[u][r], w, acc[r]);
                }
            }
        }
        __syncthreads();
        if (jg > 0) {
#pragma unroll
            for (int r = 0; r < BC; ++r) part[jg - 1][r][col] = acc[r];
        }
        __syncthreads();
        if (jg == 0 && kok) {
#pragma unroll
            for (int r = 0; r < BC; ++r) {
                const int b = b0 + r;
                if (b >= g.B) break;
                float v = acc[r];
#pragma unroll
                for (int q = 0; q < NN_JG - 1; ++q) v += part[q][r][col];
                if (P.gate) v = P.gate[(size_t)b * P.ldg + k] > 0.f ? v * P.gate_scale : 0.f;
                if (P.add) v += P.add[(size_t)b * P.lda + k];
                float* o = P.out + (size_t)b * P.ldo + k;
                *o = P.accumulate ? *o + v : v;
            }
        }
    }
}

// weight gradient: dW[j, k..k+3] += sum_b dy[b, j] in[b, k..k+3];  dbias[j] += sum_b dy[b, j]
struct OutP { const float* dy; int lddy; const float* in; int ldin; float* dW; int ldw; float* dbias; int J, K; int blk0; };
struct OutGrp { int n, B, total_blk; OutP p[8]; };
__global__ __launch_bounds__(TNT) void tail_outer_kernel(const OutGrp g) {
    int bid = blockIdx.x, pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if (bid >= g.p[i].blk0) pi = i;
    const OutP& P = g.p[pi];
    bid -= P.blk0;
    const int k4 = (P.K + 3) >> 2;
    const long idx = (long)bid * TNT + threadIdx.x;
    if (idx >= (long)P.J * k4) return;
    const int j = (int)(idx / k4), k = (int)(idx % k4) * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, sb = 0.f;
    const bool full = k + 3 < P.K && ((P.ldin | P.ldw) & 3) == 0 && (((uintptr_t)P.in | (uintptr_t)P.dW) & 15) == 0;
    for (int b = 0; b < g.B; ++b) {
        const float dv = P.dy[(size_t)b * P.lddy + j];
        const float* ip = P.in + (size_t)b * P.ldin + k;
        if (full) {
            const f32x4 v = *(const f32x4*)ip;
            a0 = fmaf(dv, v[0], a0); a1 = fmaf(dv, v[1], a1); a2 = fmaf(dv, v[2], a2); a3 = fmaf(dv, v[3], a3);
        } else {
            a0 = fmaf(dv, ip[0], a0);
            if (k + 1 < P.K) a1 = fmaf(dv, ip[1], a1);
            if (k + 2 < P.K) a2 = fmaf(dv, ip[2], a2);
            if (k + 3 < P.K) a3 = fmaf(dv, ip[3], a3);
        }
        sb += dv;
    }
    float* o = P.dW + (size_t)j * P.ldw + k;
    if (full) {
        f32x4 v = *(f32x4*)o;
        v[0] += a0; v[1] += a1; v[2] += a2; v[3] += a3;
        *(f32x4*)o = v;
    } else {
        o[0] += a0;
        if (k + 1 < P.K) o[1] += a1;
        if (k + 2 < P.K) o[2] += a2;
        if (k + 3 < P.K) o[3] += a3;
    }
    if (P.dbias && k == 0) P.dbias[j] += sb;
}

// dzp = (dh t + dz) z (1 - z),  dtp = dh z (1 - t^2)      ([B, n d])
__global__ void tail_gmu_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ z, const float* __restrict__ t,
                                    const float* __restrict__ dz, float* __restrict__ dzp, float* __restrict__ dtp, int B, int d, int n) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n * d) return;
    const int b = idx / (n * d), c = idx % d;
    const float g = dh[(size_t)b * d + c], zz = z[idx], tt = t[idx];
    dzp[idx] = (g * tt + (dz ? dz[idx] : 0.f)) * zz * (1.f - zz);      // dz: gradient of the returned gates, usually absent
    dtp[idx] = g * zz * (1.f - tt * tt);
}

// rows 0 and N_i - 1 of d(top_i), d(mid_i) = dx[:, i d : (i+1) d]  (+= when the row is both: N_i = 1);  dextra = dx[:, 3 d :]
struct UnpickP { float* dtop[3]; float* dmid[3]; float* dextra; const float* dx; int N[3]; int B, d, n; };
__global__ void tail_unpick_kernel(const UnpickP P) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nd = P.n * P.d;
    if (idx >= P.B * nd) return;
    const int b = idx / nd, q = idx % nd, i = q / P.d, c = q % P.d;
    const float v = P.dx[idx];
    if (i < 3) {
        const size_t r0 = (size_t)b * P.d + c, r1 = ((size_t)(P.N[i] - 1) * P.B + b) * P.d + c;
        const float w = P.N[i] == 1 ? 2.f * v : v;
        P.dtop[i][r0] = w; P.dmid[i][r0] = w;
        if (P.N[i] > 1) { P.dtop[i][r1] = v; P.dmid[i][r1] = v; }
    } else if (P.dextra) {
        P.dextra[(size_t)b * P.d + c] = v;
    }
}

int check_desc(const bpm_tail_desc* t) {
    if (!t || t->B < 1 || t->d < 1 || (t->n != 3 && t->n != 4) || t->C < 1) return BPM_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (!t->top[i] || !t->mid[i] || t->N[i] < 1) return BPM_ERR_ARG;
    if (t->n == 4 && !t->extra) return BPM_ERR_ARG;
    for (int i = 0; i < t->n; ++i)
        if (!t->Wh[i] || !t->Wg[i]) return BPM_ERR_ARG;
    if (!t->W1 || !t->b1 || !t->W2 || !t->b2 || !t->Wo || !t->bo) return BPM_ERR_ARG;
    if (!t->x || !t->z || !t->t || !t->h || !t->p1 || !t->y || !t->logits) return BPM_ERR_ARG;
    return 0;
}

int launch_linear(LinGrp& g, hipStream_t s) {
    int col = 0;
    for (int i = 0; i < g.n; ++i) { g.p[i].col0 = col; col += g.p[i].N; }
    g.total_cols = col;
    hipLaunchKernelGGL(tail_linear_kernel, dim3((col + TNT / 64 - 1) / (TNT / 64)), dim3(TNT), 0, s, g);
    BPM_CHECK_LAUNCH();
    return 0;
}
int launch_nn(NnGrp& g, hipStream_t s) {
    int blk = 0;
    for (int i = 0; i < g.n; ++i) { g.p[i].blk0 = blk; blk += (g.p[i].K + 63) / 64; }
    g.total_blk = blk;
    hipLaunchKernelGGL(tail_nn_kernel, dim3(blk), dim3(64 * NN_JG), 0, s, g);
    BPM_CHECK_LAUNCH();
    return 0;
}
int launch_outer(OutGrp& g, hipStream_t s) {
    int blk = 0;
    for (int i = 0; i < g.n; ++i) {
        g.p[i].blk0 = blk;
        blk += (int)(((long)g.p[i].J * ((g.p[i].K + 3) / 4) + TNT - 1) / TNT);
    }
    g.total_blk = blk;
    hipLaunchKernelGGL(tail_outer_kernel, dim3(blk), dim3(TNT), 0, s, g);
    BPM_CHECK_LAUNCH();
    return 0;
}
LinP lin(const float* W, int ldw, const float* in, int ldin, const float* bias, float* out, int ldo, int N, int K, int act) {
    LinP p{};
    p.W = W; p.ldw = ldw; p.in = in; p.ldin = ldin; p.bias = bias; p.out = out; p.ldo = ldo; p.N = N; p.K = K; p.act = act;
    p.drop = bpm_make_drop(0.f, 0, 0);
    return p;
}

}  // namespace

extern "C" int bpm_tail_fwd(const bpm_tail_desc* t, uint64_t seed, void* stream) {
    int rc = check_desc(t);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int B = t->B, d = t->d, n = t->n, nd = n * d;
    PickP pk{};
    for (int i = 0; i < 3; ++i) { pk.top[i] = t->top[i]; pk.mid[i] = t->mid[i]; pk.N[i] = t->N[i]; }
    pk.extra = t->extra; pk.x = t->x; pk.B = B; pk.d = d; pk.n = n;
    hipLaunchKernelGGL(tail_pick_kernel, dim3((B * nd + TNT - 1) / TNT), dim3(TNT), 0, s, pk);
    BPM_CHECK_LAUNCH();
    LinGrp g{};
    g.B = B; g.n = 2 * n;
    for (int i = 0; i < n; ++i) {
        g.p[i] = lin(t->Wg[i], nd, t->x, nd, nullptr, t->z + i * d, nd, d, nd, ACT_SIGMOID);          // z_i = sigmoid(G_i cat)
        g.p[n + i] = lin(t->Wh[i], d, t->x + i * d, nd, nullptr, t->t + i * d, nd, d, d, ACT_TANH);   // t_i = tanh(W_i x_i)
    }
    if ((rc = launch_linear(g, s))) return rc;
    hipLaunchKernelGGL(tail_combine_kernel, dim3((B * d + TNT - 1) / TNT), dim3(TNT), 0, s, t->z, t->t, t->h, B, d, n);
    BPM_CHECK_LAUNCH();
    g.n = 1;
    g.p[0] = lin(t->W1, d, t->h, d, t->b1, t->p1, d, d, d, ACT_RELU_DROP);
    g.p[0].drop = bpm_make_drop(t->out_dropout, seed, t->drop_site);
    g.seedp = bpm_seed_ptr(seed);
    if ((rc = launch_linear(g, s))) return rc;
    g.seedp = nullptr;
    g.p[0] = lin(t->W2, d, t->p1, d, t->b2, t->y, d, d, d, ACT_NONE);
    g.p[0].resid = t->h; g.p[0].ldr = d;
    if ((rc = launch_linear(g, s))) return rc;
    g.p[0] = lin(t->Wo, d, t->y, d, t->bo, t->logits, t->C, t->C, d, ACT_NONE);
    return launch_linear(g, s);
}

extern "C" int bpm_tail_bwd(const bpm_tail_desc* t, const bpm_tail_grads* q, void* stream) {
    int rc = check_desc(t);
    if (rc) return rc;
    if (!q || !q->dlogits || !q->dy || !q->dp1 || !q->dh || !q->dzp || !q->dtp || !q->dx) return BPM_ERR_ARG;
    for (int i = 0; i < t->n; ++i)
        if (!q->dWh[i] || !q->dWg[i]) return BPM_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (!q->dtop[i] || !q->dmid[i]) return BPM_ERR_ARG;
    if (!q->dW1 || !q->db1 || !q->dW2 || !q->db2 || !q->dWo || !q->dbo) return BPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int B = t->B, d = t->d, n = t->n, nd = n * d, Cn = t->C;
    NnGrp g{};
    g.B = B; g.n = 1;
    auto nn = [&](const float* dy, int lddy, const float* W, int ldw, int J, int K, float* out, int ldo, int acc) {
        NnP p{};
        p.dy = dy; p.lddy = lddy; p.W = W; p.ldw = ldw; p.J = J; p.K = K; p.out = out; p.ldo = ldo; p.accumulate = acc;
        p.gate_scale = 1.f;
        return p;
    };
    OutGrp o{};
    o.B = B; o.n = 1;
    auto outer = [&](const float* dy, int lddy, const float* in, int ldin, float* dW, int ldw, float* db, int J, int K) {
        OutP p{};
        p.dy = dy; p.lddy = lddy; p.in = in; p.ldin = ldin; p.dW = dW; p.ldw = ldw; p.dbias = db; p.J = J; p.K = K;
        return p;
    };
    // out_layer
    g.p[0] = nn(q->dlogits, Cn, t->Wo, d, Cn, d, q->dy, d, 0);
    if ((rc = launch_nn(g, s))) return rc;
    o.p[0] = outer(q->dlogits, Cn, t->y, d, q->dWo, d, q->dbo, Cn, d);
    if ((rc = launch_outer(o, s))) return rc;
    // proj2 (+ residual): d(p1 pre-activation) = (dy W2) * relu'(.) * dropout multiplier: p1 > 0 exactly where both kept
    g.p[0] = nn(q->dy, d, t->W2, d, d, d, q->dp1, d, 0);
    g.p[0].gate = t->p1; g.p[0].ldg = d;
    g.p[0].gate_scale = t->out_dropout > 0.f ? 1.f / (1.f - t->out_dropout) : 1.f;
    if ((rc = launch_nn(g, s))) return rc;
    o.p[0] = outer(q->dy, d, t->p1, d, q->dW2, d, q->db2, d, d);
    if ((rc = launch_outer(o, s))) return rc;
    // proj1: dh = dy + dp1 W1
    g.p[0] = nn(q->dp1, d, t->W1, d, d, d, q->dh, d, 0);
    g.p[0].add = q->dy; g.p[0].lda = d;
    if ((rc = launch_nn(g, s))) return rc;
    o.p[0] = outer(q->dp1, d, t->h, d, q->dW1, d, q->db1, d, d);
    if ((rc = launch_outer(o, s))) return rc;
    // gated fusion
    hipLaunchKernelGGL(tail_gmu_bwd_kernel, dim3((B * nd + TNT - 1) / TNT), dim3(TNT), 0, s, q->dh, t->z, t->t, q->dz, q->dzp, q->dtp, B, d, n);
    BPM_CHECK_LAUNCH();
    // dx = sum_i dzp_i G_i  (full width, one launch per gate so that each launch has one owner per element) ...
    for (int i = 0; i < n; ++i) {
        g.p[0] = nn(q->dzp + i * d, nd, t->Wg[i], nd, d, nd, q->dx, nd, i > 0);
        if ((rc = launch_nn(g, s))) return rc;
    }
    // ... + dtp_i W_i into column block i (disjoint blocks: one grouped launch)
    g.n = n;
    for (int i = 0; i < n; ++i) g.p[i] = nn(q->dtp + i * d, nd, t->Wh[i], d, d, d, q->dx + i * d, nd, 1);
    if ((rc = launch_nn(g, s))) return rc;
    o.n = 2 * n;
    for (int i = 0; i < n; ++i) {
        o.p[i] = outer(q->dzp + i * d, nd, t->x, nd, q->dWg[i], nd, nullptr, d, nd);
        o.p[n + i] = outer(q->dtp + i * d, nd, t->x + i * d, nd, q->dWh[i], d, nullptr, d, d);
    }
    if ((rc = launch_outer(o, s))) return rc;
    UnpickP u{};
    for (int i = 0; i < 3; ++i) { u.dtop[i] = q->dtop[i]; u.dmid[i] = q->dmid[i]; u.N[i] = t->N[i]; }
    u.dextra = q->dextra; u.dx = q->dx; u.B = B; u.d = d; u.n = n;
    hipLaunchKernelGGL(tail_unpick_kernel, dim3((B * nd + TNT - 1) / TNT), dim3(TNT), 0, s, u);
    BPM_CHECK_LAUNCH();
    return 0;
}
