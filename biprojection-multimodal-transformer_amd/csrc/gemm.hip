// Grouped MFMA GEMM for the BPMulT hot path (gfx950).
//
// One launch computes up to BPM_MAX_GROUP independent problems (the six
// crossmodal encoders of a level run in lock-step), each
//   C[M,N] = epilogue( sum_k X(m,k) * Y(n,k) )
// in three operand arrangements
//   NT: X = A[M,K] (k contiguous), Y = W[N,K] (k contiguous)   forward  y = x W^T
//   NN: X = A[M,K] (k contiguous), Y = B[K,N] (n contiguous)   dgrad    dx = dy W
//   TN: X = A[K,M] (m contiguous), Y = B[K,N] (n contiguous)   wgrad    dW = dy^T x
// Operands are CT (= float: exact f32 MFMA 16x16x4, or bf16: MFMA 16x16x32,
// f32 accumulate) in row-major buffers whose leading dimension is a multiple
// of 32 elements with zero padding, so tiles are moved in whole 16-byte
// chunks with chunk-granular guards only.
//
// Tile: 128(M) x 64(N) per 256-thread workgroup, 4 waves as 2x2, each wave
// 64x32 = 4x2 MFMA tiles.  k is consumed one or two 64-byte k-steps per stage
// through double-buffered, register-staged LDS images:
//   k-contiguous operand  -> image [rows][stage bytes + 16 B pad]  read with ds_read_b128
//   k-strided   operand  -> image [k][rows*sz + pad]        read transposed
//                            (ds_read_b64_tr_b16 for bf16, ds_read_b32 for f32)
// The MFMA is issued "swapped" (Y rows as the A operand, X rows as the B
// operand) so that a lane owns 4 consecutive n of one output row m and the
// epilogue stores 16 B (f32) / 8 B (bf16) vectors.
//
// Epilogue (all optional, per problem): + bias[n], + bias[m], * alpha, ReLU,
// gate by (aux > 0) * s (ReLU/dropout backward), dropout, + residual, then
// store as f32 row-major (optionally += or atomicAdd for split-K), CT
// row-major (pad columns zeroed) or CT head-major [B,H,T,dhp] (attention
// operand layout).
#include "bpm_common.h"
#include "bpm_prof.h"
#include "../../include/bpmult_hip.h"

namespace {

#ifndef BPM_GEMM_BM
#define BPM_GEMM_BM 128
#endif
#ifndef BPM_GEMM_BN
#define BPM_GEMM_BN 64
#endif
#ifndef BPM_GEMM_WM
#define BPM_GEMM_WM 2
#endif
#ifndef BPM_GEMM_WN
#define BPM_GEMM_WN 2
#endif
constexpr int BM = BPM_GEMM_BM, BN = BPM_GEMM_BN, WM = BPM_GEMM_WM, WN = BPM_GEMM_WN;
constexpr int NTHREADS = 64 * WM * WN;
constexpr int TM = BM / WM / 16;          // 4 MFMA tiles along m per wave
constexpr int TN = BN / WN / 16;          // 2 along n
// 64-byte k-steps per LDS stage.  Measured on MI355X at the model's shapes (tools/bench_kernels.py):
// the forward / dgrad GEMMs have short k loops (K = 300) and are latency bound, so the smaller stage
// (27 KB of LDS, 5 workgroups per CU in flight) wins by 25-50 %; the weight-gradient GEMM (K = T*B rows)
// prefers the longer stage.
constexpr int KS_FWD = 1, KS_WGRAD = 2;

struct Prob {
    const char* X; const char* Y; char* C;
    int M, N, K;
    int ldx, ldy, ldc;
    const float* bias_n; const float* bias_m;
    const float* resid; int ldr;
    const char* gate; int ldg; float gate_scale;
    float alpha;
    DropCfg drop;
    float* colsum;
    int flags;
    int out_kind;
    int hB, hH, hT, hdh, hdhp;
    int tile0, tiles_m, tiles_n, splitk;
};

struct Group {
    int nprob;
    int total_tiles;
    Prob p[BPM_MAX_GROUP];
};

// PERM (bf16, k-contiguous, one k-step per stage): the other operand of this product is read TRANSPOSED with the
// conflict-free "ctile" row assignment (lane group g takes k = 4g..4g+3 and 16+4g..16+4g+3), so this image stores
// the eight 8-byte halves of a 64-byte k-step re-ordered: half h -> chunk (h & 3), slot (h >> 2).  One
// ds_read_b128 then yields the same k order as the transposed read of the other side.
template <typename CT, bool KCONTIG, int ROWS, int KSTEPS, bool PERM = false>
struct Side {
    static constexpr int BKB = KSTEPS * 64;          // bytes of k per stage and row
    // k-contiguous image.  One k-step per stage: rows of exactly 64 B with the 16-byte chunk index XOR-swizzled
    // by swz(row): ds_read_b128 serves lanes in groups {0-3,12-15,20-27},{4-11,16-19,28-31},.. (MI355X_MICROARCH
    // section LDS), i.e. 16 rows with the chunk alternating between g and g^1; swz = (-(row>>2))&3 puts those 16
    // addresses on 16 distinct 16-byte slots of the 256-byte bank row (a +16 B row pad does not: measured
    // SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).  Longer stages keep the padded rows.
    static constexpr bool SWZ = (KSTEPS == 1);
    static constexpr int ROW_STRIDE = SWZ ? 64 : BKB + 16;
    static BPM_DEV int swz(int row) { return (-(row >> 2)) & 3; }
    static constexpr int SZ = sizeof(CT);
    static constexpr int EPC = Tr<CT>::EPC;
    static constexpr int BK = KSTEPS * Tr<CT>::KSTEP;                    // elements of k per stage
    static constexpr int STRIDE = KCONTIG ? ROW_STRIDE : ROWS * SZ + Tr<CT>::TR_PAD_B;
    static constexpr int IMG_BYTES = KCONTIG ? ROWS * ROW_STRIDE : BK * STRIDE;
    static constexpr int NCHUNK = ROWS * BKB / 16;
    static constexpr int PER_THREAD = (NCHUNK + NTHREADS - 1) / NTHREADS;

    // global -> registers.  rows_bound: valid rows of this side (M or N);
    // k_lo/k_hi: contraction range of this block; ld: leading dim (elements).
    static BPM_DEV void load(const char* base, int ld, int row0, int rows_bound, int k0, int k_hi, int kext,
                             u32x4 (&reg)[PER_THREAD], int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NTHREADS;
            bool ok;
            size_t off;
            if (KCONTIG) {
                const int row = c / (BKB / 16), kc = c % (BKB / 16);
                const int k = k0 + kc * EPC;
                ok = (row0 + row < rows_bound) && (k < k_hi) && (k < kext);
                off = ((size_t)(row0 + row) * ld + k) * SZ;
            } else {
                constexpr int CPR = ROWS * SZ / 16;
                const int kr = c / CPR, cc = c % CPR;
                const int col = row0 + cc * EPC;
                ok = (k0 + kr < k_hi) && (col + EPC <= ld) && (col < rows_bound);
                off = ((size_t)(k0 + kr) * ld + col) * SZ;
            }
            if (NCHUNK % NTHREADS) ok = ok && (c < NCHUNK);
            reg[i] = ok ? *(const u32x4*)(base + off) : u32x4{0u, 0u, 0u, 0u};
        }
    }
    static BPM_DEV void store(char* img, const u32x4 (&reg)[PER_THREAD], int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NTHREADS;
            if ((NCHUNK % NTHREADS) && c >= NCHUNK) continue;
            int dst;
            if (KCONTIG) {
                const int row = c / (BKB / 16), kc = c % (BKB / 16);
                if constexpr (PERM && sizeof(CT) == 2) {
                    static_assert(!PERM || KSTEPS == 1, "PERM images hold one k-step");
                    const int h0 = 2 * kc, h1 = 2 * kc + 1, sw = swz(row);
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    *(u32x2*)(img + row * 64 + (((h0 & 3) ^ sw) << 4) + ((h0 >> 2) << 3)) = u32x2{reg[i][0], reg[i][1]};
                    *(u32x2*)(img + row * 64 + (((h1 & 3) ^ sw) << 4) + ((h1 >> 2) << 3)) = u32x2{reg[i][2], reg[i][3]};
                    continue;
                }
                dst = row * ROW_STRIDE + (SWZ ? (kc ^ swz(row)) : kc) * 16;
            } else {
                constexpr int CPR = ROWS * SZ / 16;
                const int kr = c / CPR, cc = c % CPR;
                dst = kr * STRIDE + cc * 16;
            }
            *(u32x4*)(img + dst) = reg[i];
        }
    }
    // operand chunk for the 16 rows starting at r0, k-step ks of the stage
    static BPM_DEV typename Tr<CT>::frag frag(const char* img, int r0, int ks, int lane) {
        if (KCONTIG) {
            if (SWZ) {
                const int r = lane & 15, g = lane >> 4;
                return *(const typename Tr<CT>::frag*)(img + (r0 + r) * 64 + ((g ^ swz(r0 + r)) << 4));
            }
            return read_rowfrag<CT>(img, ROW_STRIDE, r0, ks, lane);
        }
        // transposed read with the "ctile" row assignment: a 32-lane half touches 8 consecutive k rows, which
        // with a row of 4*odd 8-byte units (TR_PAD_B) is bank-conflict free (the 8g+j assignment is 2-way)
        return Tr<CT>::read_tr(img, STRIDE, ks * Tr<CT>::KSTEP, r0, lane, Tr<CT>::TR_CTILE);
    }
};

template <typename CT>
BPM_DEV void store_ct4(char* C, size_t off_elems, const float (&v)[4], int nvalid) {
    CT* p = (CT*)C + off_elems;
    if (nvalid == 4 && ((off_elems & 3) == 0)) {
        if constexpr (sizeof(CT) == 4) {
            *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
            *(bf16x4*)p = o;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) p[r] = Tr<CT>::from_f(v[r]);
    }
}

template <typename CT, bool XK, bool YK, int KSTEPS>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const Group grp) {
    typedef Side<CT, XK, BM, KSTEPS, XK && !YK> SX;     // NN: X is k-contiguous beside a transposed-read Y
    typedef Side<CT, YK, BN, KSTEPS, false> SY;
    constexpr int STAGE = SX::IMG_BYTES + SY::IMG_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // block -> (problem, split, tile)
    int bid = blockIdx.x;
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.nprob; ++i)
        if (bid >= grp.p[i].tile0) pi = i;
    const Prob& P = grp.p[pi];
    bid -= P.tile0;
    const int tiles = P.tiles_m * P.tiles_n;
    const int split = bid / tiles;
    const int t = bid % tiles;
    const int m0 = (t / P.tiles_n) * BM, n0 = (t % P.tiles_n) * BN;

    constexpr int BK = SX::BK;
    const int nkt_all = (P.K + BK - 1) / BK;
    const int per = (nkt_all + P.splitk - 1) / P.splitk;
    const int kt_lo = split * per;
    const int kt_hi = min(nkt_all, kt_lo + per);
    const int kext = (P.K + Tr<CT>::EPC - 1) / Tr<CT>::EPC * Tr<CT>::EPC;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 rx[SX::PER_THREAD], ry[SY::PER_THREAD];
    if (kt_lo < kt_hi) {
        SX::load(P.X, P.ldx, m0, P.M, kt_lo * BK, P.K, kext, rx, tid);
        SY::load(P.Y, P.ldy, n0, P.N, kt_lo * BK, P.K, kext, ry, tid);
        SX::store(smem, rx, tid);
        SY::store(smem + SX::IMG_BYTES, ry, tid);
    }
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const bool more = kt + 1 < kt_hi;
#ifdef BPM_EXP_NOLOAD
        if (false) {
#else
        if (more) {
#endif
            SX::load(P.X, P.ldx, m0, P.M, (kt + 1) * BK, P.K, kext, rx, tid);
            SY::load(P.Y, P.ldy, n0, P.N, (kt + 1) * BK, P.K, kext, ry, tid);
        }
        const char* ix = smem + cur * STAGE;
        const char* iy = ix + SX::IMG_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            typename Tr<CT>::frag fx[TM], fy[TN];
#pragma unroll
            for (int b = 0; b < TM; ++b) fx[b] = SX::frag(ix, wm * (BM / WM) + 16 * b, ks, lane);
#pragma unroll
            for (int a = 0; a < TN; ++a) fy[a] = SY::frag(iy, wn * (BN / WN) + 16 * a, ks, lane);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
#ifndef BPM_EXP_NOMMA
                    acc[a][b] = Tr<CT>::mma(fy[a], fx[b], acc[a][b]);
#else
                    asm volatile("" :: "v"(fy[a]), "v"(fx[b]));
#endif
                }
        }
        if (more) {
            char* nx = smem + (cur ^ 1) * STAGE;
            SX::store(nx, rx, tid);
            SY::store(nx + SX::IMG_BYTES, ry, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---------------- epilogue ----------------
#ifdef BPM_EXP_NOEPI
    if (acc[0][0][0] == 12345.678f) ((float*)P.C)[0] = acc[0][0][1];
    return;
#endif
    const int r = lane & 15, g = lane >> 4;
    const bool lead = (split == 0);
    const bool atomic = (P.flags & BPM_GEMM_ATOMIC) != 0;
    const bool accum = (P.flags & BPM_GEMM_ACCUM) != 0;
    const bool relu = (P.flags & BPM_GEMM_RELU) != 0;
    if (kt_lo >= kt_hi && !lead) return;
    float csum[TN][4];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) csum[a][q] = 0.f;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + wm * (BM / WM) + 16 * b + r;
        if (m >= P.M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int nb = n0 + wn * (BN / WN) + 16 * a + 4 * g;
            float v[4];
            const bool full = nb + 3 < P.N;
            // side operands: one 16-byte (f32) / 8-byte (bf16) load per lane where the 4 columns are in range and aligned
            float bn[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, gt[4] = {1.f, 1.f, 1.f, 1.f};
            if (lead && P.bias_n) {
                const float* bp = P.bias_n + nb;
                if (full && (((uintptr_t)bp & 15) == 0)) { const f32x4 t = *(const f32x4*)bp; bn[0] = t[0]; bn[1] = t[1]; bn[2] = t[2]; bn[3] = t[3]; }
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (nb + q < P.N) bn[q] = bp[q];
                }
            }
            if (lead && P.resid) {
                const float* rp = P.resid + (size_t)m * P.ldr + nb;
                if (full && (((uintptr_t)rp & 15) == 0)) { const f32x4 t = *(const f32x4*)rp; rs[0] = t[0]; rs[1] = t[1]; rs[2] = t[2]; rs[3] = t[3]; }
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (nb + q < P.N) rs[q] = rp[q];
                }
            }
            if (P.gate) {
                const CT* gp = (const CT*)P.gate + (size_t)m * P.ldg + nb;
                if (nb + 3 < P.ldg) {          // gate rows are padded CT rows: an aligned 4-element read is always in bounds
                    if constexpr (sizeof(CT) == 4) { const f32x4 t = *(const f32x4*)gp; gt[0] = t[0]; gt[1] = t[1]; gt[2] = t[2]; gt[3] = t[3]; }
                    else { const bf16x4 t = *(const bf16x4*)gp; gt[0] = (float)t[0]; gt[1] = (float)t[1]; gt[2] = (float)t[2]; gt[3] = (float)t[3]; }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (nb + q < P.N) gt[q] = Tr<CT>::to_f(gp[q]);
                }
            }
            const float bm = (lead && P.bias_m) ? P.bias_m[m] : 0.f;
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (P.drop.thresh != 0) {
                const uint32_t i0 = (uint32_t)m * (uint32_t)P.N + (uint32_t)nb;
                if ((i0 & 1u) == 0) {                       // (nb, nb+1) and (nb+2, nb+3) are hash pairs
                    bpm_drop_mult2(P.drop, i0, dm[0], dm[1]);
                    bpm_drop_mult2(P.drop, i0 + 2, dm[2], dm[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dm[q] = bpm_drop_mult(P.drop, i0 + q);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nb + q;
                float x = 0.f;
                if (n < P.N) {
                    x = (acc[a][b][q] + bn[q] + bm) * P.alpha;
                    if (relu) x = fmaxf(x, 0.f);
                    if (P.gate) x = gt[q] > 0.f ? x * P.gate_scale : 0.f;
                    x *= dm[q];
                    csum[a][q] += x;
                    x += rs[q];
                }
                v[q] = x;
            }
#ifdef BPM_EXP_NOSTORE
            if (v[0] + v[1] + v[2] + v[3] == 12345.678f) ((float*)P.C)[0] = v[0];
            continue;
#endif
            if (P.out_kind == BPM_OUT_F32) {
                float* c = (float*)P.C + (size_t)m * P.ldc + nb;
                if (full && !atomic && (((uintptr_t)c & 15) == 0)) {
                    f32x4 o = f32x4{v[0], v[1], v[2], v[3]};
                    if (accum) o += *(const f32x4*)c;
                    *(f32x4*)c = o;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (nb + q >= P.N) continue;
                        if (atomic) atomicAdd(c + q, v[q]);
                        else if (accum) c[q] += v[q];
                        else c[q] = v[q];
                    }
                }
            } else if (P.out_kind == BPM_OUT_CT) {
                const int nvalid = min(4, P.ldc - nb);   // pad columns [N, ldc) get zeros
                if (nvalid > 0) store_ct4<CT>(P.C, (size_t)m * P.ldc + nb, v, nvalid);
            } else {  // BPM_OUT_HEADS: m = t*B + b, n = h*dh + c  ->  [B,H,T,dhp]
                const int tt = m / P.hB, bb = m % P.hB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nb + q;
                    if (n >= P.N) continue;
                    const int h = n / P.hdh, c = n % P.hdh;
                    ((CT*)P.C)[(((size_t)bb * P.hH + h) * P.hT + tt) * P.hdhp + c] = Tr<CT>::from_f(v[q]);
                }
            }
        }
    }
    if (P.colsum) {   // uniform per block: every lane takes part in the shuffles
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = csum[a][q];
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                const int n = n0 + wn * (BN / WN) + 16 * a + 4 * g + q;
                if (r == 0 && n < P.N) atomicAdd(P.colsum + n, v);
            }
    }
}

template <typename CT>
int launch(int variant, const Group& g, hipStream_t s) {
    dim3 grid(g.total_tiles), block(NTHREADS);
    switch (variant) {
        case BPM_GEMM_NT: hipLaunchKernelGGL((gemm_kernel<CT, true, true, KS_FWD>), grid, block, 0, s, g); break;
        case BPM_GEMM_NN: hipLaunchKernelGGL((gemm_kernel<CT, true, false, KS_FWD>), grid, block, 0, s, g); break;
        case BPM_GEMM_TN: hipLaunchKernelGGL((gemm_kernel<CT, false, false, KS_WGRAD>), grid, block, 0, s, g); break;
        default: return BPM_ERR_ARG;
    }
    BPM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int bpm_gemm_grouped(int dtype, int variant, const bpm_gemm_problem* probs, int nprob, uint64_t seed, void* stream) {
    if (nprob < 1 || nprob > BPM_MAX_GROUP || !probs) return BPM_ERR_ARG;
    const int sz = dtype == BPM_BF16 ? 2 : 4;
    Group g;
    g.nprob = nprob;
    int tile = 0;
    for (int i = 0; i < nprob; ++i) {
        const bpm_gemm_problem& q = probs[i];
        Prob& p = g.p[i];
        if (q.M < 1 || q.N < 1 || q.K < 1 || !q.A || !q.B || !q.C) return BPM_ERR_ARG;
        // operand leading dims must keep every row 16-byte aligned
        if ((q.lda * sz) % 16 || (q.ldb * sz) % 16) return BPM_ERR_ALIGN;
        if (((uintptr_t)q.A | (uintptr_t)q.B) & 15) return BPM_ERR_ALIGN;
        p.X = (const char*)q.A; p.Y = (const char*)q.B; p.C = (char*)q.C;
        p.M = q.M; p.N = q.N; p.K = q.K;
        p.ldx = q.lda; p.ldy = q.ldb; p.ldc = q.ldc;
        p.bias_n = q.bias_n; p.bias_m = q.bias_m;
        p.resid = q.resid; p.ldr = q.ldr;
        p.gate = (const char*)q.gate; p.ldg = q.ldg; p.gate_scale = q.gate_scale;
        p.alpha = q.alpha;
        p.drop = bpm_make_drop(q.drop_p, seed, q.drop_site);
        p.colsum = q.colsum;
        p.flags = q.flags; p.out_kind = q.out_kind;
        p.hB = q.heads_B; p.hH = q.heads_H; p.hT = q.heads_T; p.hdh = q.heads_dh; p.hdhp = q.heads_dhp;
        if (q.out_kind == BPM_OUT_HEADS && (q.heads_B < 1 || q.heads_dh < 1 || q.heads_H * q.heads_dh != q.N)) return BPM_ERR_ARG;
        p.tiles_m = (q.M + BM - 1) / BM;
        p.tiles_n = (q.N + BN - 1) / BN;
        if (q.out_kind == BPM_OUT_CT && p.tiles_n * BN < q.ldc) return BPM_ERR_ARG;
        p.splitk = q.splitk > 1 ? q.splitk : 1;
        if (p.splitk > 1 && !((q.flags & BPM_GEMM_ATOMIC) && q.out_kind == BPM_OUT_F32)) return BPM_ERR_ARG;
        p.tile0 = tile;
        tile += p.tiles_m * p.tiles_n * p.splitk;
    }
    g.total_tiles = tile;
    hipStream_t s = (hipStream_t)stream;
    double flops = 0;
    for (int i = 0; i < nprob; ++i) flops += 2.0 * probs[i].M * (double)probs[i].N * probs[i].K;
    BpmProfScope prof(BPM_K_GEMM_NT + variant, s, flops);
    return dtype == BPM_BF16 ? launch<bf16_t>(variant, g, s) : launch<float>(variant, g, s);
}
