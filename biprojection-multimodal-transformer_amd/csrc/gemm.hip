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
// of 32 elements with zero padding, so tiles move in whole 16-byte chunks.
//
// Two kernels share one epilogue:
//
// * gemm_tiled_kernel -- the general one.  128(M) x 64(N) tile per 256-thread
//   workgroup (4 waves as 2x2, each 64x32 = 4x2 MFMA tiles), k consumed one or
//   two 64-byte k-steps per stage through double-buffered, register-staged LDS
//   images: k-contiguous operands as 64-byte rows with an XOR swizzle that is
//   conflict free for ds_read_b128's lane groups, k-strided operands as
//   [k][rows] images read transposed (ds_read_b64_tr_b16 / ds_read_b32).
//
// * gemm_dma_kernel (gemm_dma.h) -- bf16 products with k extents of whole 128-byte stages and at least
//   one full 128 x 128 tile (hidden >= 512: every encoder GEMM at hidden 768 / 1536).  128 x 128 or 256 x 128
//   tile, each wave 64 x 64, operands moved global -> LDS by `buffer_load ... lds` into a ring of stages that
//   stays in flight across barriers (counted vmcnt), 64 k per stage.
//
// The MFMA is issued "swapped" (Y rows as the A operand, X rows as the B
// operand) so that a lane owns 4 consecutive n of one output row m and the
// epilogue moves 16 B (f32) / 8 B (bf16) vectors.
//
// Epilogue (all optional, per problem): + bias[n], + bias[m], * alpha, ReLU,
// gate by (aux > 0) * s (ReLU/dropout backward), dropout, column sums (bias
// gradients), + residual, then store as f32 row-major (optionally += ; float
// atomics only for split-K), CT row-major (pad columns zeroed) or CT
// head-major [B,H,T,dhp] (attention operand layout).
#include <cstdlib>
#include <type_traits>

#include "bpm_common.h"
#include "bpm_prof.h"
#include "../../include/bpmult_hip.h"

namespace {

constexpr int BM = 128, BN = 64, WM = 2, WN = 2;   // tiled kernel: 128 x 64 workgroup tile, 2 x 2 waves
constexpr int NTHREADS = 64 * WM * WN;
constexpr int TM = BM / WM / 16;          // 4 MFMA tiles along m per wave
constexpr int TN = BN / WN / 16;          // 2 along n
// 64-byte k-steps per LDS stage of the tiled kernel.  Measured on MI355X at the model's shapes: the
// forward / dgrad GEMMs with short k loops are latency bound and the smaller stage (12 KB, 5 workgroups
// per CU) wins by 25-50 %; the weight-gradient GEMM (K = T*B rows) prefers the longer stage.
// waves per SIMD the tiled kernel is compiled for (register budget 512 / BPM_TILED_MINW per lane): 5 fits without
// spilling since the epilogue no longer keeps a per-row offset table (84-88 VGPRs); 6 spills and runs 1.3-1.8x slower.
// Measured: 4 and 5 perform the same (the kernel is not occupancy-bound), so the extra wave is free headroom.
constexpr int BPM_TILED_MINW = 5;
// s_setprio(1) around the MFMA cluster: alone the GEMMs gain 1-10 % (wgrad 85 -> 80 us), inside the training step they
// then take issue slots from the critical-path kernels of the other stream and the step gets SLOWER (17.0 -> 17.4 ms)
constexpr int BPM_SETPRIO = 0;
// issue priority of critical-path kernels (forward / dgrad GEMMs not flagged BACKGROUND, attention, LayerNorm) over
// the side stream's: 16.92 -> 16.63 ms/step
constexpr int BPM_BASE_PRIO = 1;
constexpr int BPM_DEEP_TN = 2;
constexpr int BPM_DEEP_FWD = 1;
constexpr int KS_FWD = 1, KS_WGRAD = 2;
struct Prob {                              // 168 bytes: 24 of them (+ the header) are one 4 KB kernel argument
    const char* X; const char* Y; char* C;
    int M, N, K;
    int ldx, ldy, ldc;
    const float* bias_n; const float* bias_m;
    const float* resid; const char* gate;
    int ldr, ldg; float gate_scale;
    float alpha;
    float* colsum;
    float* colsum_x;      // TN: [M] += sum_k X[k, m]
    DropCfg drop;
    int flags;
    int out_kind;
    int hB, hH, hT, hdh, hdhp;
    int tile0, tiles_m, tiles_n, splitk;
};
static_assert(sizeof(Prob) == 168, "Prob layout");

struct Group {
    const uint64_t* seedp;     // dropout seed read at execution time (BPM_SEED_INDIRECT), or nullptr
    int nprob;
    int total_tiles;
    Prob p[BPM_GEMM_MAX_GROUP];
};
static_assert(sizeof(Group) + 16 <= 4096, "the problem table travels as a kernel argument (4 KB)");

BPM_DEV int swz4(int row) { return (-(row >> 2)) & 3; }

// ---------------------------------------------------------------------------
// shared epilogue: one 16x16 accumulator tile (lane: row m, columns nb..nb+3)
// ---------------------------------------------------------------------------
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

template <typename CT>
BPM_DEV void epilogue_tile(const Prob& P, const DropCfg& drop, bool lead, int m, int nb, const f32x4& acc, float (&csum)[4]) {
    if (m >= P.M) return;
    const bool atomic = (P.flags & BPM_GEMM_ATOMIC) != 0;
    const bool accum = (P.flags & BPM_GEMM_ACCUM) != 0;
    const bool relu = (P.flags & BPM_GEMM_RELU) != 0;
    const bool full = nb + 3 < P.N;
    float v[4];
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
    if (drop.thresh != 0) {
        const uint32_t i0 = (uint32_t)m * (uint32_t)P.N + (uint32_t)nb;
        if ((i0 & 3u) == 0) {                       // nb .. nb+3 is one hash quad
            bpm_drop_mult4(drop, i0, dm[0], dm[1], dm[2], dm[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) dm[q] = bpm_drop_mult(drop, i0 + q);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = 0.f;
        if (nb + q < P.N) {
            x = (acc[q] + bn[q] + bm) * P.alpha;
            if (relu) x = fmaxf(x, 0.f);
            if (P.gate) x = gt[q] > 0.f ? x * P.gate_scale : 0.f;
            x *= dm[q];
            csum[q] += x;
            x += rs[q];
        }
        v[q] = x;
    }
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
        const int nvalid = min(4, ((P.flags & BPM_GEMM_CT_NARROW) ? P.N : P.ldc) - nb);   // pad columns [N, ldc) get zeros
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

// column sums of a wave's tiles: reduce over the 16 row lanes, one atomic per column
BPM_DEV void flush_colsum(const Prob& P, float (&csum)[4], int nb, int r) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = csum[q];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (r == 0 && nb + q < P.N) atomicAdd(P.colsum + nb + q, v);
    }
}

// ---------------------------------------------------------------------------
// fast epilogue.  The general epilogue_tile above costs ~25 VALU instructions per output element (64-bit
// address arithmetic per side operand, per-lane alignment tests that diverge, per-element bound tests and two
// integer divisions per element for the head-major scatter) -- more than the MFMA time of a K = 300 product.
// When a problem satisfies wave-uniform preconditions (N % 4 == 0, 16-byte aligned side operands with
// leading dimensions % 4 == 0, no split-K / atomics / row bias) the same arithmetic is done with the
// row-invariant part hoisted per output row and straight-line 4-wide code per tile.
// ---------------------------------------------------------------------------
BPM_DEV bool epi_fast_ok(const Prob& P) {
    const uintptr_t al = (uintptr_t)P.bias_n | (uintptr_t)P.resid | (uintptr_t)P.C | (uintptr_t)P.gate;
    return (P.N & 3) == 0 && (al & 15) == 0 && ((P.ldr | P.ldc | P.ldg) & 3) == 0 && !P.bias_m &&
           !(P.flags & BPM_GEMM_ATOMIC) && (P.splitk == 1 || (P.flags & BPM_GEMM_BATCHED)) && !(P.resid && (P.flags & BPM_GEMM_ACCUM));
}

struct EpiRow {            // per output row m: everything that does not depend on the column
    bool ok;
    uint32_t offc, offr, offg, didx;   // m*ldc, m*ldr, m*ldg, m*N
    uint32_t hrow;                     // head-major: (b*H*T + t) * dhp
};

BPM_DEV EpiRow epi_row(const Prob& P, int m) {
    EpiRow e;
    e.ok = m < P.M;
    const uint32_t um = (uint32_t)(e.ok ? m : 0);
    e.offc = um * (uint32_t)P.ldc; e.offr = um * (uint32_t)P.ldr; e.offg = um * (uint32_t)P.ldg; e.didx = um * (uint32_t)P.N;
    e.hrow = 0;
    if (P.out_kind == BPM_OUT_HEADS) {
        const uint32_t tt = um / (uint32_t)P.hB, bb = um - tt * (uint32_t)P.hB;
        e.hrow = (bb * (uint32_t)(P.hH * P.hT) + tt) * (uint32_t)P.hdhp;
    }
    return e;
}

// Fast epilogue of one wave's column block: rows (m0 + 16*b + r), b < NB, columns nb..nb+3.
// Two phases.  Phase 1 issues EVERY side-operand load of the NB tiles back to back (bias, gate, residual or the
// previous value for +=) from clamped, always-valid addresses; phase 2 does the arithmetic and the stores.
// Interleaving them per tile (load, compute, store, load, ...) serialises one memory round trip per tile, because
// the loads may alias the stores and cannot be hoisted by the compiler: measured 6.6 us of a 16 us workgroup.
template <typename CT, int NB>
struct EpiSide {                                        // side operands of NB tiles, loaded ahead of the arithmetic
    typedef typename std::conditional<sizeof(CT) == 4, f32x4, bf16x4>::type gate_t;
    f32x4 bias;
    f32x4 addv[NB];
    gate_t gt[NB];
};

template <typename CT, int NB>
BPM_DEV void epi_fast_load(const Prob& P, int mrow, int nb, EpiSide<CT, NB>& s) {
    // rows mrow + 16*b: their offsets are recomputed where needed (a few multiplies) instead of living in a 24-register
    // table through the whole epilogue -- the table was what pushed the kernel past the 4 -> 5 waves/SIMD budget
    typedef typename EpiSide<CT, NB>::gate_t gate_t;
    const uint32_t nbc = nb < P.N ? (uint32_t)nb : 0u;  // clamped column for the loads
    const bool accum = P.out_kind == BPM_OUT_F32 && (P.flags & BPM_GEMM_ACCUM);
    s.bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (P.bias_n) s.bias = *(const f32x4*)(P.bias_n + nbc);
    if (P.gate) {
#pragma unroll
        for (int b = 0; b < NB; ++b) s.gt[b] = *(const gate_t*)((const CT*)P.gate + (uint32_t)min(mrow + 16 * b, P.M - 1) * (uint32_t)P.ldg + nbc);
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) s.gt[b] = gate_t{};
    }
    if (P.resid) {
#pragma unroll
        for (int b = 0; b < NB; ++b) s.addv[b] = *(const f32x4*)(P.resid + (uint32_t)min(mrow + 16 * b, P.M - 1) * (uint32_t)P.ldr + nbc);
    } else if (accum) {
#pragma unroll
        for (int b = 0; b < NB; ++b) s.addv[b] = *(const f32x4*)((const float*)P.C + (uint32_t)min(mrow + 16 * b, P.M - 1) * (uint32_t)P.ldc + nbc);
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b) s.addv[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <typename CT, int NB>
BPM_DEV void epi_fast_apply(const Prob& P, const DropCfg& drop, int mrow, int nb, const f32x4 (&acc)[NB], const EpiSide<CT, NB>& s, f32x4& csum,
                            uint32_t coff = 0) {     // coff: element offset of this batch element's output (BPM_GEMM_BATCHED)
    const bool colok = nb < P.N;                        // N % 4 == 0: a lane's 4 columns are all in or all out
    const bool f32out = P.out_kind == BPM_OUT_F32;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        EpiRow e = epi_row(P, mrow + 16 * b);
        e.offc += coff;
        const bool valid = e.ok && colok;
        f32x4 x = acc[b];
        if (valid) {
            x += s.bias;
            if (P.alpha != 1.f) x *= P.alpha;
            if (P.flags & BPM_GEMM_RELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = fmaxf(x[q], 0.f);
            }
            if (P.gate) {
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = (float)s.gt[b][q] > 0.f ? x[q] * P.gate_scale : 0.f;
            }
            if (drop.thresh != 0) {                     // m*N + nb is a multiple of 4: one hash quad
                float d0, d1, d2, d3;
                bpm_drop_mult4(drop, e.didx + (uint32_t)nb, d0, d1, d2, d3);
                x[0] *= d0; x[1] *= d1; x[2] *= d2; x[3] *= d3;
            }
            csum += x;
            x += s.addv[b];
        }
        if (f32out) {
            if (valid) *(f32x4*)((float*)P.C + e.offc + nb) = x;
        } else if (P.out_kind == BPM_OUT_CT) {
            if (e.ok && nb < ((P.flags & BPM_GEMM_CT_NARROW) ? P.N : P.ldc)) {   // pad columns [N, ldc) receive zeros
                if (!valid) x = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (sizeof(CT) == 4) *(f32x4*)((float*)P.C + e.offc + nb) = x;
                else { bf16x4 o; o[0] = (bf16_t)x[0]; o[1] = (bf16_t)x[1]; o[2] = (bf16_t)x[2]; o[3] = (bf16_t)x[3]; *(bf16x4*)((bf16_t*)P.C + e.offc + nb) = o; }
            }
        } else if (valid) {                             // head-major: one division per tile, heads advance by carry
            uint32_t h = (uint32_t)nb / (uint32_t)P.hdh, c = (uint32_t)nb - h * (uint32_t)P.hdh;
            const uint32_t hstride = (uint32_t)(P.hT * P.hdhp);
            CT* base = (CT*)P.C + e.hrow;
            if (sizeof(CT) == 2 && (P.hdh & 3) == 0) {  // the 4 columns stay inside one head
                bf16x4 o; o[0] = (bf16_t)x[0]; o[1] = (bf16_t)x[1]; o[2] = (bf16_t)x[2]; o[3] = (bf16_t)x[3];
                *(bf16x4*)((bf16_t*)base + h * hstride + c) = o;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    base[h * hstride + c] = Tr<CT>::from_f(x[q]);
                    if (++c == (uint32_t)P.hdh) { c = 0; ++h; }
                }
            }
        }
    }
}

template <typename CT, int NB>
BPM_DEV void epilogue_fast(const Prob& P, const DropCfg& drop, int mrow, int nb, const f32x4 (&acc)[NB], f32x4& csum, uint32_t coff = 0) {
    EpiSide<CT, NB> s;
    epi_fast_load<CT, NB>(P, mrow, nb, s);
    epi_fast_apply<CT, NB>(P, drop, mrow, nb, acc, s, csum, coff);
}

// one wave's column block: rows (m0 + 16*b + r), b < NB, columns nb..nb+3
template <typename CT, int NB>
BPM_DEV void epilogue_cols(const Prob& P, const DropCfg& drop, bool fast, bool lead, int m0, int r, int nb, const f32x4 (&acc)[NB], const EpiRow (&rows)[NB],
                           uint32_t coff = 0) {
    if (fast) {
        f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
        epilogue_fast<CT, NB>(P, drop, m0 + r, nb, acc, cs, coff);
        if (P.colsum) { float c4[4] = {cs[0], cs[1], cs[2], cs[3]}; flush_colsum(P, c4, nb, r); }
    } else {
        float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NB; ++b) epilogue_tile<CT>(P, drop, lead, m0 + 16 * b + r, nb, acc[b], csum);
        if (P.colsum) flush_colsum(P, csum, nb, r);
    }
}

// ---------------------------------------------------------------------------
// LDS images of the tiled kernel
// ---------------------------------------------------------------------------
// PERM (bf16, k-contiguous, one k-step per stage): the other operand of this product is read TRANSPOSED with the
// conflict-free "ctile" row assignment (lane group g takes k = 4g..4g+3 and 16+4g..16+4g+3), so this image stores
// the eight 8-byte halves of a 64-byte k-step re-ordered: half h -> chunk (h & 3), slot (h >> 2).  One
// ds_read_b128 then yields the same k order as the transposed read of the other side.
template <typename CT, bool KCONTIG, int ROWS, int KSTEPS, bool PERM = false>
struct Side {
    static constexpr int BKB = KSTEPS * 64;          // bytes of k per stage and row
    // k-contiguous image.  One k-step per stage: rows of exactly 64 B with the 16-byte chunk index XOR-swizzled
    // by swz4(row): ds_read_b128 serves lanes in groups {0-3,12-15,20-27},{4-11,16-19,28-31},.. (MI355X_MICROARCH
    // section LDS), i.e. 16 rows with the chunk alternating between g and g^1; swz4 = (-(row>>2))&3 puts those 16
    // addresses on 16 distinct 16-byte slots of the 256-byte bank row (a +16 B row pad does not: measured
    // SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).  Longer stages keep the padded rows.
    static constexpr bool SWZ = (KSTEPS == 1);
    static constexpr int ROW_STRIDE = SWZ ? 64 : BKB + 16;
    static constexpr int SZ = sizeof(CT);
    static constexpr int EPC = Tr<CT>::EPC;
    static constexpr int BK = KSTEPS * Tr<CT>::KSTEP;                    // elements of k per stage
    static constexpr int STRIDE = KCONTIG ? ROW_STRIDE : ROWS * SZ + Tr<CT>::TR_PAD_B;
    static constexpr int IMG_BYTES = KCONTIG ? ROWS * ROW_STRIDE : BK * STRIDE;
    static constexpr int NCHUNK = ROWS * BKB / 16;
    static constexpr int PER_THREAD = (NCHUNK + NTHREADS - 1) / NTHREADS;

    // global -> registers.  rows_bound: valid rows of this side (M or N); k_hi: contraction bound; ld: leading dim.
    // Loads are UNCONDITIONAL from a clamped offset (0 = the matrix origin) and the validity of each chunk comes
    // back as a bit mask that store() applies: a guarded load (`ok ? *p : 0`) compiles to a branch around the
    // load plus a zero fill of its destination, and the waitcnt pass then drains every outstanding load at the
    // loop head -- which serialises the prefetch pipeline.
    static BPM_DEV uint32_t load(const char* base, int ld, int row0, int rows_bound, int k0, int k_hi,
                                 u32x4 (&reg)[PER_THREAD], int tid) {
        uint32_t mask = 0;
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NTHREADS;
            bool ok;
            size_t off;
            if (KCONTIG) {
                const int row = c / (BKB / 16), kc = c % (BKB / 16);
                const int k = k0 + kc * EPC;
                ok = (row0 + row < rows_bound) && (k < k_hi);
                off = ((size_t)(row0 + row) * ld + k) * SZ;
            } else {
                constexpr int CPR = ROWS * SZ / 16;
                const int kr = c / CPR, cc = c % CPR;
                const int col = row0 + cc * EPC;
                ok = (k0 + kr < k_hi) && (col + EPC <= ld) && (col < rows_bound);
                off = ((size_t)(k0 + kr) * ld + col) * SZ;
            }
            if (NCHUNK % NTHREADS) ok = ok && (c < NCHUNK);
            reg[i] = *(const u32x4*)(base + (ok ? off : (size_t)0));
            mask |= ok ? (1u << i) : 0u;
        }
        return mask;
    }
    static BPM_DEV void store(char* img, const u32x4 (&reg)[PER_THREAD], uint32_t mask, int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NTHREADS;
            if ((NCHUNK % NTHREADS) && c >= NCHUNK) continue;
            const u32x4 v = ((mask >> i) & 1u) ? reg[i] : u32x4{0u, 0u, 0u, 0u};
            int dst;
            if (KCONTIG) {
                const int row = c / (BKB / 16), kc = c % (BKB / 16);
                if constexpr (PERM && sizeof(CT) == 2) {
                    static_assert(!PERM || KSTEPS == 1, "PERM images hold one k-step");
                    const int h0 = 2 * kc, h1 = 2 * kc + 1, sw = swz4(row);
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    *(u32x2*)(img + row * 64 + (((h0 & 3) ^ sw) << 4) + ((h0 >> 2) << 3)) = u32x2{v[0], v[1]};
                    *(u32x2*)(img + row * 64 + (((h1 & 3) ^ sw) << 4) + ((h1 >> 2) << 3)) = u32x2{v[2], v[3]};
                    continue;
                }
                dst = row * ROW_STRIDE + (SWZ ? (kc ^ swz4(row)) : kc) * 16;
            } else {
                constexpr int CPR = ROWS * SZ / 16;
                const int kr = c / CPR, cc = c % CPR;
                dst = kr * STRIDE + cc * 16;
            }
            *(u32x4*)(img + dst) = v;
        }
    }
    // Hardware-bounded loader (FAST): one 32-bit byte offset per chunk, computed once; the k position of a stage
    // goes into the scalar offset of the buffer load and the range check of the buffer descriptor returns zeros
    // past the end of the matrix (rows >= M of a k-contiguous operand, k >= K of a k-strided one).  Needs the
    // caller's BPM_GEMM_KPAD_ZERO promise for the k tail of k-contiguous rows.  No per-iteration address or mask
    // arithmetic is left in the k loop (it was ~2/3 of the loop's VALU instructions).
    static BPM_DEV void voffsets(int ld, int row0, int tid, int (&voff)[PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int c = tid + i * NTHREADS;
            int off;
            if (KCONTIG) {
                const int row = c / (BKB / 16), kc = c % (BKB / 16);
                off = (row0 + row) * ld * SZ + kc * 16;
            } else {
                constexpr int CPR = ROWS * SZ / 16;
                const int kr = c / CPR, cc = c % CPR;
                off = (kr * ld + row0 + cc * EPC) * SZ;
            }
            voff[i] = ((NCHUNK % NTHREADS) && c >= NCHUNK) ? 0x7FFFFFF0 : off;
        }
    }
    static BPM_DEV int stage_step(int ld) { return KCONTIG ? BKB : BK * ld * SZ; }     // bytes between k stages
    static BPM_DEV void load_fast(__amdgpu_buffer_rsrc_t rsrc, const int (&voff)[PER_THREAD], int soff, u32x4 (&reg)[PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) reg[i] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], soff, 0);
    }
    // operand chunk for the 16 rows starting at r0, k-step ks of the stage
    static BPM_DEV typename Tr<CT>::frag frag(const char* img, int r0, int ks, int lane) {
        if (KCONTIG) {
            if (SWZ) {
                const int r = lane & 15, g = lane >> 4;
                return *(const typename Tr<CT>::frag*)(img + (r0 + r) * 64 + ((g ^ swz4(r0 + r)) << 4));
            }
            return read_rowfrag<CT>(img, ROW_STRIDE, r0, ks, lane);
        }
        // transposed read with the "ctile" row assignment: a 32-lane half touches 8 consecutive k rows, which
        // with a row of 4*odd 8-byte units (TR_PAD_B) is bank-conflict free (the 8g+j assignment is 2-way)
        return Tr<CT>::read_tr(img, STRIDE, ks * Tr<CT>::KSTEP, r0, lane, Tr<CT>::TR_CTILE);
    }
};

BPM_DEV const Prob& pick_problem(const Group& grp, int& bid) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.nprob; ++i)
        if (bid >= grp.p[i].tile0) pi = i;
    bid -= grp.p[pi].tile0;
    return grp.p[pi];
}

// ---------------------------------------------------------------------------
// general tiled kernel
// ---------------------------------------------------------------------------
#ifdef BPM_GEMM_TRACE
// debug build only: per-workgroup timestamps (100 MHz realtime counter) of the tiled kernel's phases
__device__ unsigned long long g_trace[8192 * 16];
#define BPM_TRACE(slot) do { if (threadIdx.x == 0 && blockIdx.x < 8192 && (slot) < 16) \
        g_trace[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BPM_TRACE(slot) do { } while (0)
#endif

template <typename CT, bool XK, bool YK, int KSTEPS, bool DEEP = false, bool FAST = false, int BMT = BM>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu((!XK && !YK) ? (BMT == 64 ? 6 : 4) : BPM_TILED_MINW, (!XK && !YK) ? (BMT == 64 ? 6 : 4) : BPM_TILED_MINW))) void gemm_tiled_kernel(const Group grp) {
    typedef Side<CT, XK, BMT, KSTEPS, XK && !YK> SX;
    constexpr int TMT = BMT / WM / 16;                  // MFMA tiles along m per wave (BMT = 128 or 64 rows per workgroup)     // NN: X is k-contiguous beside a transposed-read Y
    typedef Side<CT, YK, BN, KSTEPS, false> SY;
    constexpr int STAGE = SX::IMG_BYTES + SY::IMG_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    BPM_TRACE(0);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const Prob& P = pick_problem(grp, bid);
    if (BPM_BASE_PRIO && XK && !(P.flags & BPM_GEMM_BACKGROUND)) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);   // critical path
    const int tiles = P.tiles_m * P.tiles_n;
    const int split = bid / tiles;                      // split-K slice -- or the batch element of a BPM_GEMM_BATCHED problem
    const int t = bid % tiles;
    const int m0 = (t / P.tiles_n) * BMT, n0 = (t % P.tiles_n) * BN;
    const bool batched = (P.flags & BPM_GEMM_BATCHED) != 0;      // uniform; element strides travel in hB / hH / hT

    constexpr int BK = SX::BK;
    const int nkt_all = (P.K + BK - 1) / BK;
    const int per = batched ? nkt_all : (nkt_all + P.splitk - 1) / P.splitk;
    const int kt_lo = batched ? 0 : split * per;
    const int kt_hi = min(nkt_all, kt_lo + per);

    f32x4 acc[TN][TMT];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TMT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // TN only: column sums of X (bias gradient beside a weight gradient) in the workgroups of the first N tile: one MFMA
    // per X fragment against an operand of ones, the m tiles of a wave row dealt over its WN waves (one wave doing all of
    // them has 50 % more MFMAs than the rest and the workgroup's barriers make everyone wait for it)
    constexpr int XBT = (TMT + WN - 1) / WN;
    [[maybe_unused]] f32x4 xs[XBT];
    [[maybe_unused]] bool do_xs = false;
    if constexpr (!XK && !YK) {
        do_xs = P.colsum_x != nullptr && n0 == 0;                  // wave-uniform
#pragma unroll
        for (int b = 0; b < XBT; ++b) xs[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto compute = [&](const char* ix) {
        const char* iy = ix + SX::IMG_BYTES;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            typename Tr<CT>::frag fx[TMT], fy[TN];
#pragma unroll
            for (int b = 0; b < TMT; ++b) fx[b] = SX::frag(ix, wm * (BMT / WM) + 16 * b, ks, lane);
            if constexpr (!XK && !YK) {
                if (do_xs) {
                    typename Tr<CT>::frag one;
#pragma unroll
                    for (int j = 0; j < Tr<CT>::EPC; ++j) one[j] = Tr<CT>::from_f(1.0f);
#pragma unroll
                    for (int b = 0; b < TMT; ++b)
                        if (b / XBT == wn) xs[b % XBT] = Tr<CT>::mma(one, fx[b], xs[b % XBT]);
                }
            }
#pragma unroll
            for (int a = 0; a < TN; ++a) fy[a] = SY::frag(iy, wn * (BN / WN) + 16 * a, ks, lane);
            if (BPM_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TMT; ++b) acc[a][b] = Tr<CT>::mma(fy[a], fx[b], acc[a][b]);
            if (BPM_SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
    };

    const char* const Xp = P.X + (batched ? (size_t)split * (size_t)P.hB * sizeof(CT) : (size_t)0);
    const char* const Yp = P.Y + (batched ? (size_t)split * (size_t)P.hH * sizeof(CT) : (size_t)0);
    const int ldx = P.ldx, ldy = P.ldy, Mb = P.M, Nb = P.N, Kb = P.K;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsx, rsy;
    [[maybe_unused]] int vx[SX::PER_THREAD], vy[SY::PER_THREAD];
    [[maybe_unused]] int stepx = 0, stepy = 0;
    if constexpr (FAST) {
        // tight descriptors (see desc_bytes in gemm_dma.h): whole k stages of a k-contiguous row, whole chunks of a k-strided one
        constexpr int E = Tr<CT>::EPC;
        const int kceil = (Kb + BK - 1) / BK * BK;
        // (an operand whose rows overlap, BPM_GEMM_x_OVERLAP, owns its last row's full length past the leading dimension)
        const int wx = XK ? kceil : (Mb + E - 1) / E * E, wy = YK ? kceil : (Nb + E - 1) / E * E;
        rsx = __builtin_amdgcn_make_buffer_rsrc((void*)Xp, 0, ((XK ? Mb : Kb) - 1) * ldx * (int)sizeof(CT) + ((P.flags & BPM_GEMM_A_OVERLAP) ? wx : min(wx, ldx)) * (int)sizeof(CT), 0x00020000);
        rsy = __builtin_amdgcn_make_buffer_rsrc((void*)Yp, 0, ((YK ? Nb : Kb) - 1) * ldy * (int)sizeof(CT) + ((P.flags & BPM_GEMM_B_OVERLAP) ? wy : min(wy, ldy)) * (int)sizeof(CT), 0x00020000);
        SX::voffsets(ldx, m0, tid, vx);
        SY::voffsets(ldy, n0, tid, vy);
        stepx = SX::stage_step(ldx);
        stepy = SY::stage_step(ldy);
    }
    // stage kt -> registers; returns nothing, masks by reference (all ones on the hardware-bounded path)
    auto LD = [&](int kt, u32x4 (&rx_)[SX::PER_THREAD], u32x4 (&ry_)[SY::PER_THREAD], uint32_t& mx_, uint32_t& my_) {
        if constexpr (FAST) {
            SX::load_fast(rsx, vx, kt * stepx, rx_);
            SY::load_fast(rsy, vy, kt * stepy, ry_);
            mx_ = 0xFFFFFFFFu; my_ = 0xFFFFFFFFu;
        } else {
            mx_ = SX::load(Xp, ldx, m0, Mb, kt * BK, Kb, rx_, tid);
            my_ = SY::load(Yp, ldy, n0, Nb, kt * BK, Kb, ry_, tid);
        }
    };
    if constexpr (DEEP) {
        // Two register sets: the loads of k-stage kt+2 are issued before the arithmetic of stage kt, so two
        // stages are in flight per workgroup and a stage's L2 round trip is spread over two iterations.
        u32x4 rx0[SX::PER_THREAD], ry0[SY::PER_THREAD], rx1[SX::PER_THREAD], ry1[SY::PER_THREAD];
        uint32_t mx0 = 0, my0 = 0, mx1 = 0, my1 = 0;
        auto ld0 = [&](int kt) { LD(kt, rx0, ry0, mx0, my0); };
        auto ld1 = [&](int kt) { LD(kt, rx1, ry1, mx1, my1); };
        auto st0 = [&](char* img) { SX::store(img, rx0, mx0, tid); SY::store(img + SX::IMG_BYTES, ry0, my0, tid); };
        auto st1 = [&](char* img) { SX::store(img, rx1, mx1, tid); SY::store(img + SX::IMG_BYTES, ry1, my1, tid); };
        // stage kt lives in set (kt - kt_lo) & 1
        // loads past kt_hi are issued too (every chunk masked off, address clamped to the origin): a load inside a
        // conditional makes the waitcnt pass assume the shorter queue and drain the newer stage with the older one
        ld0(kt_lo);
        st0(smem);
        ld1(kt_lo + 1);
        __syncthreads();
        BPM_TRACE(1);
        int cur = 0;
#pragma unroll 1
        for (int kt = kt_lo; kt < kt_hi; kt += 2) {
            // even step: LDS holds kt, set1 carries kt+1, set0 is free -> kt+2
            ld0(kt + 2);
            compute(smem + cur * STAGE);
            if (kt + 1 < kt_hi) st1(smem + (cur ^ 1) * STAGE);
            __syncthreads();
            BPM_TRACE(2 + kt - kt_lo);
            cur ^= 1;
            if (kt + 1 >= kt_hi) break;
            // odd step: LDS holds kt+1, set0 carries kt+2, set1 is free -> kt+3
            ld1(kt + 3);
            compute(smem + cur * STAGE);
            if (kt + 2 < kt_hi) st0(smem + (cur ^ 1) * STAGE);
            __syncthreads();
            BPM_TRACE(3 + kt - kt_lo);
            cur ^= 1;
        }
    } else {
        u32x4 rx[SX::PER_THREAD], ry[SY::PER_THREAD];
        uint32_t mx = 0, my = 0;
        LD(kt_lo, rx, ry, mx, my);
        SX::store(smem, rx, mx, tid);
        SY::store(smem + SX::IMG_BYTES, ry, my, tid);
        __syncthreads();
        BPM_TRACE(1);
        int cur = 0;
#pragma unroll 1
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            const bool more = kt + 1 < kt_hi;
            LD(kt + 1, rx, ry, mx, my);                                       // unconditional (masked off past K)
            compute(smem + cur * STAGE);
            if (more) {
                char* nx = smem + (cur ^ 1) * STAGE;
                SX::store(nx, rx, mx, tid);
                SY::store(nx + SX::IMG_BYTES, ry, my, tid);
            }
            __syncthreads();
            BPM_TRACE(2 + kt - kt_lo);
            cur ^= 1;
        }
    }

    const int r = lane & 15, g = lane >> 4;
    const bool lead = batched || (split == 0);
    if constexpr (!XK && !YK) {
        if (do_xs && g == 0) {               // every row of the ones-product holds the sums: lane r has column m
#pragma unroll
            for (int b = 0; b < TMT; ++b) {
                const int m = m0 + wm * (BMT / WM) + 16 * b + r;
                if (b / XBT == wn && m < P.M) P.colsum_x[m] += xs[b % XBT][0];
            }
        }
    }
    if (kt_lo >= kt_hi && !lead) return;
    const bool fast = epi_fast_ok(P);                   // wave-uniform
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const int mw = m0 + wm * (BMT / WM);
    EpiRow rows[TMT];
#pragma unroll
    for (int b = 0; b < TMT; ++b) rows[b] = epi_row(P, mw + 16 * b + r);
    const uint32_t coff = batched ? (uint32_t)split * (uint32_t)P.hT : 0u;    // (host: 4-wide epilogue only, no side operands)
    BPM_TRACE(12);
    // (explicitly unrolled: a rolled loop would index the accumulators dynamically and send them to scratch)
    auto epi = [&](auto A) {           // colsum shuffles: uniform per workgroup, every lane takes part
        constexpr int a = decltype(A)::value;
        if constexpr (a < TN) {
            epilogue_cols<CT, TMT>(P, drop, fast, lead, mw, r, n0 + wn * (BN / WN) + 16 * a + 4 * g, acc[a], rows, coff);
            BPM_TRACE(13 + a);
        }
    };
    epi(std::integral_constant<int, 0>{}); epi(std::integral_constant<int, 1>{});
    epi(std::integral_constant<int, 2>{}); epi(std::integral_constant<int, 3>{});
    static_assert(TN <= 4, "epilogue unrolled for up to 4 column tiles per wave");
#ifdef BPM_GEMM_TRACE
    __builtin_amdgcn_s_waitcnt(0);     // stores issued; vmcnt drained
    BPM_TRACE(15);
#endif
}

// ---------------------------------------------------------------------------
// skinny kernel: problems of at most 16 rows (the query side of the level-2 encoders under dead-row elimination: rows
// {0, N-1} x batch -- SURVEY A.10).  Such a product is a weight-streaming operation (1.2 MB of weights against 25 KB of
// activations at hidden 768) and the 128 x 64 kernel spent it on latency: one workgroup per 64 columns walking all of K
// through LDS stages with a barrier each (28-37 us per launch for ~3 us of memory traffic; 95-105 us with fp32 operands).
// Here a workgroup still owns 16 rows x 64 columns, but its four waves split K between them, every wave loads its MFMA
// operands straight from global memory (k-contiguous X and, for NT, W rows: 16 bytes per lane, all loads of a chunk of
// k-steps in flight before the first MFMA; for NN the [k][n] weight slice goes through a wave-private LDS block and is
// read back transposed), and the four partial tiles are summed through LDS before the shared epilogue.
// ---------------------------------------------------------------------------
constexpr int SK_CHUNK = 6;            // k-steps (64 bytes of k each) whose loads are in flight together per wave

template <typename CT, bool YK>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const Group grp) {
    typedef typename Tr<CT>::frag frag;
    constexpr int SZ = sizeof(CT), KS = Tr<CT>::KSTEP, EPC = Tr<CT>::EPC;
    constexpr int YSTRIDE = 64 * SZ + Tr<CT>::TR_PAD_B;          // wave-private [k][64 n] image of one k-step (NN)
    constexpr int YIMG = KS * YSTRIDE;
    __shared__ __attribute__((aligned(16))) char smem[4 * 4 * 64 * 16 > 4 * YIMG ? 4 * 4 * 64 * 16 : 4 * YIMG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const Prob& P = pick_problem(grp, bid);
    if (BPM_BASE_PRIO && !(P.flags & BPM_GEMM_BACKGROUND)) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    const int n0 = bid * 64;
    const int nks = (P.K + KS - 1) / KS;
    const int per = (nks + 3) >> 2;
    const int ks_lo = wave * per, ks_hi = min(nks, ks_lo + per);

    f32x4 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool xrow_ok = r < P.M;
    const char* xrow = P.X + (size_t)(xrow_ok ? r : 0) * P.ldx * SZ + g * 16;          // + ks * 64: this lane's chunk of k-step ks
    char* yimg = smem + wave * YIMG;

#pragma unroll 1
    for (int ks0 = ks_lo; ks0 < ks_hi; ks0 += SK_CHUNK) {
        frag fx[SK_CHUNK];
#pragma unroll
        for (int c = 0; c < SK_CHUNK; ++c) {
            const int ks = min(ks0 + c, ks_hi - 1);              // clamped: always a valid address; surplus steps are skipped below
            fx[c] = *(const frag*)(xrow + (size_t)ks * 64);
            if (!xrow_ok) fx[c] = Tr<CT>::zero();
        }
        if constexpr (YK) {
            frag fy[SK_CHUNK][4];
#pragma unroll
            for (int c = 0; c < SK_CHUNK; ++c) {
                const int ks = min(ks0 + c, ks_hi - 1);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int n = n0 + 16 * a + r;
                    fy[c][a] = *(const frag*)(P.Y + (size_t)(n < P.N ? n : 0) * P.ldy * SZ + (size_t)ks * 64 + g * 16);
                    if (n >= P.N) fy[c][a] = Tr<CT>::zero();
                }
            }
#pragma unroll
            for (int c = 0; c < SK_CHUNK; ++c)
                if (ks0 + c < ks_hi) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] = Tr<CT>::mma(fy[c][a], fx[c], acc[a]);
                }
        } else {
            // W[k][n] rows of this k-step: KS rows x 64 columns = 4 chunks of 16 bytes per lane
            constexpr int CPR = 64 * SZ / 16;                    // 16-byte chunks per k-row
            constexpr int PT = KS * CPR / 64;                    // chunks per lane and k-step
            u32x4 yv[SK_CHUNK][PT];
#pragma unroll
            for (int c = 0; c < SK_CHUNK; ++c) {
                const int ks = min(ks0 + c, ks_hi - 1);
#pragma unroll
                for (int i = 0; i < PT; ++i) {
                    const int ch = lane + 64 * i, kr = ch / CPR, cc = ch % CPR;
                    const int k = ks * KS + kr, n = n0 + cc * EPC;
                    const bool ok = k < P.K && n + EPC <= P.ldy && n < P.N;
                    yv[c][i] = *(const u32x4*)(P.Y + ((size_t)(ok ? k : 0) * P.ldy + (ok ? n : 0)) * SZ);
                    if (!ok) yv[c][i] = u32x4{0u, 0u, 0u, 0u};
                }
            }
#pragma unroll
            for (int c = 0; c < SK_CHUNK; ++c)
                if (ks0 + c < ks_hi) {                            // wave-uniform
#pragma unroll
                    for (int i = 0; i < PT; ++i) {
                        const int ch = lane + 64 * i, kr = ch / CPR, cc = ch % CPR;
                        *(u32x4*)(yimg + kr * YSTRIDE + cc * 16) = yv[c][i];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave: its LDS accesses execute in order
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        acc[a] = Tr<CT>::mma(Tr<CT>::read_tr(yimg, YSTRIDE, 0, 16 * a, lane, Tr<CT>::TR_NATURAL), fx[c], acc[a]);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next k-step overwrites the image
                }
        }
    }
    // the four waves' partial tiles -> LDS; wave w sums and finishes column tile w
    __syncthreads();
    f32x4* red = (f32x4*)smem;                                   // [wave][a][lane]
#pragma unroll
    for (int a = 0; a < 4; ++a) red[(wave * 4 + a) * 64 + lane] = acc[a];
    __syncthreads();
    f32x4 t = red[(0 * 4 + wave) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) t += red[(w * 4 + wave) * 64 + lane];
    const bool fast = epi_fast_ok(P);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    f32x4 tile[1] = {t};
    EpiRow rows[1] = {epi_row(P, r)};
    epilogue_cols<CT, 1>(P, drop, fast, true, 0, r, n0 + 16 * wave + 4 * g, tile, rows);
}

#include "gemm_dma.h"

template <typename CT>
int launch(int variant, bool fast, bool bm64, const Group& g, hipStream_t s) {
    dim3 grid(g.total_tiles), block(NTHREADS);
    if (fast) {
        switch (variant) {
            case BPM_GEMM_NT: hipLaunchKernelGGL((gemm_tiled_kernel<CT, true, true, KS_FWD, BPM_DEEP_FWD != 0, true>), grid, block, 0, s, g); break;
            case BPM_GEMM_NN: hipLaunchKernelGGL((gemm_tiled_kernel<CT, true, false, 1, BPM_DEEP_FWD != 0, true>), grid, block, 0, s, g); break;
            case BPM_GEMM_TN:
                if (bm64) hipLaunchKernelGGL((gemm_tiled_kernel<CT, false, false, 1, true, true, 64>), grid, block, 0, s, g);
                else hipLaunchKernelGGL((gemm_tiled_kernel<CT, false, false, 1, true, true>), grid, block, 0, s, g);
                break;
            default: return BPM_ERR_ARG;
        }
        BPM_CHECK_LAUNCH();
        return 0;
    }
    switch (variant) {
        case BPM_GEMM_NT: hipLaunchKernelGGL((gemm_tiled_kernel<CT, true, true, KS_FWD, BPM_DEEP_FWD != 0>), grid, block, 0, s, g); break;
        case BPM_GEMM_NN: hipLaunchKernelGGL((gemm_tiled_kernel<CT, true, false, 1, BPM_DEEP_FWD != 0>), grid, block, 0, s, g); break;
        case BPM_GEMM_TN:
            // measured (MI355X, K = 4096 rows): with at least ~2 workgroups per CU the short stage wins
            // (more resident workgroups hide the load latency); below that the long stage does
            if (g.total_tiles >= 2 * 256) hipLaunchKernelGGL((gemm_tiled_kernel<CT, false, false, 1, BPM_DEEP_TN != 0>), grid, block, 0, s, g);
            else if (BPM_DEEP_TN == 2) hipLaunchKernelGGL((gemm_tiled_kernel<CT, false, false, 1, true>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_tiled_kernel<CT, false, false, KS_WGRAD, false>), grid, block, 0, s, g);
            break;
        default: return BPM_ERR_ARG;
    }
    BPM_CHECK_LAUNCH();
    return 0;
}

// LDS-DMA kernel configurations: (waves along m, waves along n, m tiles per wave, stages); X3: split-bf16 operands
template <int WMD, int WND, int TMW, int NS, bool X3 = false, int DKT = DK>
int launch_dma_cfg(int variant, const Group& g, hipStream_t s) {
    dim3 grid(g.total_tiles), block(64 * WMD * WND);
    switch (variant) {
        case BPM_GEMM_NT: hipLaunchKernelGGL((gemm_dma_kernel<true, true, WMD, WND, TMW, NS, false, X3, DKT>), grid, block, 0, s, g); break;
        case BPM_GEMM_NN: hipLaunchKernelGGL((gemm_dma_kernel<true, false, WMD, WND, TMW, NS, false, X3, DKT>), grid, block, 0, s, g); break;
        case BPM_GEMM_TN:
            if constexpr ((16 * TMW * WMD) % 128 == 0) {
                bool xs = false;
                for (int i = 0; i < g.nprob; ++i) xs = xs || g.p[i].colsum_x != nullptr;
                if (xs) hipLaunchKernelGGL((gemm_dma_kernel<false, false, WMD, WND, TMW, NS, true, X3, DKT>), grid, block, 0, s, g);
                else hipLaunchKernelGGL((gemm_dma_kernel<false, false, WMD, WND, TMW, NS, false, X3, DKT>), grid, block, 0, s, g);
            } else return BPM_ERR_ARG;         // k-strided X: 128-column sub-images only
            break;
        default: return BPM_ERR_ARG;
    }
    BPM_CHECK_LAUNCH();
    return 0;
}

struct DmaCfg { int bm, bn; };
constexpr DmaCfg DMA_CFGS[] = {{128, 128}, {256, 128}, {256, 256}, {256, 256}, {128, 128}, {320, 256}, {256, 128}};
constexpr int CFG_TALL = 5;               // 320-row tiles (k-contiguous X only): see the tile choice in bpm_gemm_grouped
constexpr int CFG_TWO = 6;                // 256 x 128, 32-k stages, three of them: two workgroups per CU (gemm_dma.h)
constexpr int N_DMA_CFGS = sizeof(DMA_CFGS) / sizeof(DMA_CFGS[0]);

int launch_dma(int cfg, int variant, const Group& g, hipStream_t s, bool x3 = false) {
    if (x3) {                                 // the configurations the automatic choice picks (fewer instantiations)
        switch (cfg) {
            case 2: return launch_dma_cfg<2, 4, 8, 2, true>(variant, g, s);
            case 3: return launch_dma_cfg<4, 4, 4, 2, true>(variant, g, s);
            case CFG_TALL: return launch_dma_cfg<2, 4, 10, 2, true>(variant, g, s);
        }
        return BPM_ERR_ARG;
    }
    switch (cfg) {
        case 0: return launch_dma_cfg<2, 2, 4, 2>(variant, g, s);      // 128 x 128,  4 waves, 2 stages (64 KB: 2 per CU)
        case 1: return launch_dma_cfg<4, 2, 4, 2>(variant, g, s);      // 256 x 128,  8 waves
        case 2: return launch_dma_cfg<2, 4, 8, 2>(variant, g, s);      // 256 x 256,  8 waves of 128 x 64
        case 3: return launch_dma_cfg<4, 4, 4, 2>(variant, g, s);      // 256 x 256, 16 waves of 64 x 64
        case 4: return launch_dma_cfg<2, 2, 4, 3>(variant, g, s);      // 128 x 128,  3 stages
        case CFG_TALL: return launch_dma_cfg<2, 4, 10, 2>(variant, g, s);   // 320 x 256, 8 waves of 160 x 64 (NT / NN)
        case CFG_TWO: return launch_dma_cfg<4, 2, 4, 3, false, 32>(variant, g, s);   // 256 x 128, 8 waves of 64 x 64, 3 stages of 32 k: 72 KB
    }
    return BPM_ERR_ARG;
}

int num_cus() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    return ncu;
}

// tile-configuration override: -1 = automatic choice, -2 = never the LDS-DMA kernel, 0.. = force DMA_CFGS[i] where legal.
// A constant in the product library; only a -DBPM_LAB build (build/lab/libbpmult_hip_lab.so: tools/gemm_lab.py and the
// kernel tests that pin a configuration) makes it a process-global switch behind bpm_debug_gemm_force.
#ifdef BPM_LAB
int g_force_dma = -1;
#else
constexpr int g_force_dma = -1;
#endif

}  // namespace

#ifdef BPM_LAB
extern "C" int bpm_debug_gemm_force(int cfg) {
    if (cfg < -2 || cfg >= N_DMA_CFGS) return BPM_ERR_ARG;
    g_force_dma = cfg;
    return 0;
}
#endif

#ifdef BPM_GEMM_TRACE
extern "C" int bpm_debug_trace(unsigned long long* out, int nblocks) {
    if (nblocks > 8192) nblocks = 8192;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 16 * nblocks);
}
#endif

extern "C" int bpm_gemm_grouped(int dtype, int variant, const bpm_gemm_problem* probs, int nprob, uint64_t seed, void* stream) {
    if (nprob < 1 || nprob > BPM_GEMM_MAX_GROUP || !probs) return BPM_ERR_ARG;
    if (variant != BPM_GEMM_NT && variant != BPM_GEMM_NN && variant != BPM_GEMM_TN) return BPM_ERR_ARG;
    // BPM_BF16X3: A / B are split-bf16 images (bpm_split_rows: lda / ldb span the hi and the lo plane), K / M / N are those
    // of the fp32 product, CT outputs and the gate operand are fp32; only the LDS-DMA kernel computes it
    const bool x3 = dtype == BPM_BF16X3;
    if (dtype != BPM_F32 && dtype != BPM_BF16 && !x3) return BPM_ERR_ARG;
    const int sz = dtype == BPM_F32 ? 4 : 2;
    const bool xk = variant != BPM_GEMM_TN, yk = variant == BPM_GEMM_NT;
    // hardware-bounded loader: every problem promises zero k padding, k-contiguous rows are whole k stages and the
    // matrices fit 31-bit byte offsets
    bool fast = true, overlap = false;
    for (int i = 0; i < nprob && fast; ++i) {
        const bpm_gemm_problem& q = probs[i];
        const int stage_b = 64 * (variant == BPM_GEMM_NT ? KS_FWD : 1);
        const bool aov = (q.flags & BPM_GEMM_A_OVERLAP) != 0, bov = (q.flags & BPM_GEMM_B_OVERLAP) != 0;
        overlap = overlap || aov || bov;
        fast = (q.flags & BPM_GEMM_KPAD_ZERO) != 0;
        // overlapping rows: no zero padding behind a row (the next window starts there), so k must end on a stage
        if (xk) fast = fast && (aov ? q.K % 64 == 0 : ((long)q.lda * sz) % stage_b == 0 && (long)q.K <= q.lda);
        if (yk) fast = fast && (bov ? q.K % 64 == 0 : ((long)q.ldb * sz) % stage_b == 0 && (long)q.K <= q.ldb);
        const long bx = ((long)(xk ? q.M : q.K) - 1) * q.lda * sz + (long)(aov ? (xk ? q.K : q.M) : q.lda) * sz;
        const long by = ((long)(yk ? q.N : q.K) - 1) * q.ldb * sz + (long)(bov ? (yk ? q.K : q.N) : q.ldb) * sz;
        fast = fast && bx < (1l << 31) - 65536 && by < (1l << 31) - 65536;
    }
    if (overlap && !fast) return BPM_ERR_ARG;      // only the hardware-bounded loaders address overlapping rows
    // LDS-DMA kernel (gemm_dma.h): bf16, no split-K, k-contiguous operands hold whole 128-byte stages inside their
    // zero-padded rows, epilogue 4-wide.  Measured on MI355X at hidden 768, six problems of 4096 rows per launch
    // (tools/gemm_lab.py, us, register-staged 128 x 64 kernel -> the configuration chosen below): q 84 -> 55, k/v 154 ->
    // 103, out 80 -> 55, fc1 267 -> 156, fc2 245 -> 124, d(fc2) 314 -> 181, d(fc1) 185 -> 110, d(out) 82 -> 55, d(q) 63 ->
    // 47, d(k/v) 135 -> 107, FFN weight gradients 445 -> 322 (8 waves of 128 x 64); the attention weight gradients
    // (768 x 768 x 4096: nine 256 x 256 tiles per problem) stay on the 128 x 64 kernel (154 against 186).
    int dma = -1;
    if ((dtype == BPM_BF16 || x3) && fast && (g_force_dma != -2 || x3)) {
        bool legal = true, big = true;
        long tiles_tn = 0;
        for (int i = 0; i < nprob && legal; ++i) {
            const bpm_gemm_problem& q = probs[i];
            const long kceil = ((long)q.K + DK - 1) / DK * DK;
            legal = q.splitk <= 1 && !(q.flags & (BPM_GEMM_ATOMIC | BPM_GEMM_BATCHED)) && (!xk || (q.flags & BPM_GEMM_A_OVERLAP) || kceil <= q.lda) &&
                    (!yk || (q.flags & BPM_GEMM_B_OVERLAP) || kceil <= q.ldb);
            if (x3) {       // a plane (half the leading dimension) holds whole k stages / whole 128-column sub-images, no overlap
                const long pa = q.lda / 2, pb = q.ldb / 2, m128 = ((long)q.M + 127) / 128 * 128, n128 = ((long)q.N + 127) / 128 * 128;
                legal = legal && !(q.flags & (BPM_GEMM_A_OVERLAP | BPM_GEMM_B_OVERLAP)) && (q.lda & 1) == 0 && (q.ldb & 1) == 0 &&
                        (xk ? kceil <= pa : m128 <= pa) && (yk ? kceil <= pb : n128 <= pb) && (pa * 2) % 16 == 0 && (pb * 2) % 16 == 0;
            }
            // its epilogue is the 4-wide one only (epi_fast_ok, evaluated here on the host)
            const uintptr_t al = (uintptr_t)q.bias_n | (uintptr_t)q.resid | (uintptr_t)q.C | (uintptr_t)q.gate;
            legal = legal && (q.N & 3) == 0 && (al & 15) == 0 && ((q.ldr | q.ldc | q.ldg) & 3) == 0 && !q.bias_m &&
                    !(q.resid && (q.flags & BPM_GEMM_ACCUM));
            const long tm = (q.M + 255) / 256, tn = (q.N + 255) / 256;
            big = big && q.M >= 256 && q.N >= 256 && q.K >= 256 &&
                  (double)q.M * q.N >= 0.8 * (double)(tm * 256) * (double)(tn * 256);
            tiles_tn += tm * tn;
        }
        // weight gradients: one 256 x 256 tile per CU only pays when most CUs get one (measured, 768 x 768 x 4096 problems
        // with bias column sums: 24 problems = 216 tiles 227 -> 181 us, 18 = 162 tiles 169 -> 156, 12 = 108 tiles 114 -> ~150)
        if (variant == BPM_GEMM_TN && tiles_tn * 8 < num_cus() * 5 && !x3) big = false;
        // products whose columns fill 128-wide tiles but not 256-wide ones (hidden 300: 300 / 384 against 300 / 512): the
        // two-resident 256 x 128 configuration instead of the 128 x 64 kernel (lab switch BPMULT_NARROW_DMA=0)
        static const bool narrow_ok = !(std::getenv("BPMULT_NARROW_DMA") && std::atoi(std::getenv("BPMULT_NARROW_DMA")) == 0);
        bool narrow = narrow_ok && legal && !big && !x3 && variant != BPM_GEMM_TN;
        long tiles_narrow = 0;
        for (int i = 0; i < nprob && narrow; ++i) {
            const bpm_gemm_problem& q = probs[i];
            const long tm = (q.M + 255) / 256, tn = (q.N + 127) / 128;
            narrow = q.M >= 256 && q.N >= 128 && q.K >= 256 && (double)q.M * q.N >= 0.75 * (double)(tm * 256) * (double)(tn * 128);
            tiles_narrow += tm * tn;
        }
        narrow = narrow && tiles_narrow >= num_cus();
        if (x3 && !legal) return BPM_ERR_ARG;
        if (x3) big = true;                       // the caller (ops.gemm_grouped) sends what bpm_gemm_x3_eligible accepted
        if (legal && g_force_dma >= 0 && !x3) dma = g_force_dma == CFG_TALL && variant == BPM_GEMM_TN ? -1 : g_force_dma;
        else if (legal && big) {
            dma = variant == BPM_GEMM_TN ? 2 : 3;
            // Two resident workgroups per CU (256 x 128 tiles, 32-k stages; gemm_dma.h) where they measured faster than the
            // one-per-CU tiles (tools/gemm_lab.py, hidden 768, six problems of 4096 rows): the FFN weight gradients
            // 328 -> 279 us (their k loop is LDS-fill bound: a second workgroup's loads and a third stage fill the gaps);
            // twelve-problem K / V projections 104 -> 94.  Not for weight gradients that carry the bias column sums (the
            // extra accumulators spill at 128 registers: 156 -> 253 us) nor for the N = 768 products, where 320 x 256 tiles
            // fit one round (q 55 -> 59, fc2 124 -> 178).  Inside the step (bench.py, hidden 768, ms per step; BPMULT_TWO_MASK
            // is the lab switch for these classes): none 19.97-20.03, weight gradients + K / V 19.65-19.82, + fc1 (N >= 2048
            // forward products) 19.44, + d(fc2) 19.54-19.80, every N = 768 product 20.22.
            bool xs_any = false;
            long kmin = 1l << 30, kmax = 0, nmax = 0, mmax = 0;
            for (int i = 0; i < nprob; ++i) {
                xs_any = xs_any || probs[i].colsum_a != nullptr;
                kmin = probs[i].K < kmin ? probs[i].K : kmin;
                kmax = probs[i].K > kmax ? probs[i].K : kmax;
                nmax = probs[i].N > nmax ? probs[i].N : nmax;
                mmax = probs[i].M > mmax ? probs[i].M : mmax;
            }
            static const int two_mask = std::getenv("BPMULT_TWO_MASK") ? std::atoi(std::getenv("BPMULT_TWO_MASK")) : 7;
            // (hidden 1536, `bench.py --config cfg5`: the weight-gradient and fc1 classes each cost 1 % there -- 70.7 -> 71.4 / 71.6
            // ms -- so they are tied to hidden <= 1024: weight gradients of at most 1024 rows, forward products of K <= 1024)
            const bool two_tn = (two_mask & 1) && variant == BPM_GEMM_TN && !xs_any && kmin >= 1024 && mmax <= 1024 && !x3;
            bool two_kv = (two_mask & 2) && variant == BPM_GEMM_NT && nprob >= 12 && nmax <= 1024 && !x3;
            if ((two_mask & 4) && variant == BPM_GEMM_NT && nmax >= 2048 && kmax <= 1024 && !x3) two_kv = true;          // fc1
            if ((two_mask & 8) && variant == BPM_GEMM_NN && nmax >= 2048 && !x3) two_kv = true;          // d(fc2)
            if ((two_mask & 16) && variant == BPM_GEMM_TN && xs_any && !x3) two_kv = true;               // attention weight gradients
            if ((two_mask & 32) && variant != BPM_GEMM_TN && nmax <= 1024 && !x3) two_kv = true;         // every N = 768 product
            if (variant != BPM_GEMM_TN) {
                // One workgroup per CU: a launch takes ceil(tiles / CUs) rounds of one tile each, and at the model's
                // shapes the tile count sits just above a multiple of the CU count (six problems of 4096 x 768 are 288
                // tiles of 256 x 256: a second round for 32 of them).  320-row tiles (13 instead of 16 per 4096 rows: 234
                // tiles, one round of 1.25x the work) win whenever they save a round; per flop the 16-wave 256-row
                // kernel is ~5 % faster.
                const long ncu = num_cus();
                long t256 = 0, t320 = 0;
                for (int i = 0; i < nprob; ++i) {
                    const long tn = (probs[i].N + 255) / 256;
                    t256 += (probs[i].M + 255) / 256 * tn;
                    t320 += (probs[i].M + 319) / 320 * tn;
                }
                const long r256 = (t256 + ncu - 1) / ncu, r320 = (t320 + ncu - 1) / ncu;
                if (r320 * 320 * 21 < r256 * 256 * 20) dma = CFG_TALL;
            }
            if (two_tn || two_kv) dma = CFG_TWO;
        } else if (narrow) dma = CFG_TWO;
    }
    // at most 16 rows per problem (level-2 query side under dead-row elimination): the skinny kernel (see there)
    bool skinny = dma < 0 && !x3 && variant != BPM_GEMM_TN && fast;
    for (int i = 0; i < nprob && skinny; ++i) {
        const bpm_gemm_problem& q = probs[i];
        skinny = q.M <= 16 && q.splitk <= 1 && !(q.flags & (BPM_GEMM_ATOMIC | BPM_GEMM_A_OVERLAP | BPM_GEMM_B_OVERLAP | BPM_GEMM_BATCHED)) &&
                 ((long)q.lda * sz) % 64 == 0 && (variant != BPM_GEMM_NT || ((long)q.ldb * sz) % 64 == 0);
    }
    // under-filled weight-gradient launches (fewer than two 128-row workgroups per CU) run 64-row workgroups:
    // measured 98 -> 77 us for the 24 attention weight gradients of a layer (360 -> 600 workgroups)
    int bm_tile = BM, bn_tile = BN;
    if (skinny) { bm_tile = 16; bn_tile = 64; }
    if (dma >= 0) { bm_tile = DMA_CFGS[dma].bm; bn_tile = DMA_CFGS[dma].bn; }
    else if (variant == BPM_GEMM_TN && fast && BM == 128) {
        long t128 = 0;
        for (int i = 0; i < nprob; ++i)
            t128 += (long)((probs[i].M + BM - 1) / BM) * ((probs[i].N + BN - 1) / BN) * (probs[i].splitk > 1 ? probs[i].splitk : 1);
        if (t128 < 2 * 256) bm_tile = 64;
    }
    Group g;
    g.seedp = bpm_seed_ptr(seed);
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
        p.colsum_x = q.colsum_a;
        if (q.colsum_a && (variant != BPM_GEMM_TN || q.splitk > 1)) return BPM_ERR_ARG;
        p.flags = q.flags; p.out_kind = q.out_kind;
        p.hB = q.heads_B; p.hH = q.heads_H; p.hT = q.heads_T; p.hdh = q.heads_dh; p.hdhp = q.heads_dhp;
        if (q.out_kind == BPM_OUT_HEADS && (q.heads_B < 1 || q.heads_dh < 1 || q.heads_H * q.heads_dh != q.N)) return BPM_ERR_ARG;
        const int ntiles = (q.N + bn_tile - 1) / bn_tile;
        if (q.out_kind == BPM_OUT_CT && !(q.flags & BPM_GEMM_CT_NARROW) && ntiles * bn_tile < q.ldc) return BPM_ERR_ARG;
        p.tile0 = tile;
        p.tiles_m = (q.M + bm_tile - 1) / bm_tile;
        p.tiles_n = ntiles;
        p.splitk = q.splitk > 1 ? q.splitk : 1;
        if (p.splitk > 1 && !((q.flags & BPM_GEMM_ATOMIC) && q.out_kind == BPM_OUT_F32)) return BPM_ERR_ARG;
        if (q.flags & BPM_GEMM_BATCHED) {
            // `batch` independent products of one shape: operand / output i starts batch_stride_* elements behind i - 1.  The
            // 128 x 64 kernel only; the batch index travels where the split-K slice does, the strides in the head fields
            // (plain stores through the 4-wide epilogue, no side operands).
            const uintptr_t al = (uintptr_t)q.C | (uintptr_t)q.bias_n;
            if (x3 || dma >= 0 || skinny || q.batch < 1 || q.batch > 4096 || q.splitk > 1 || q.out_kind == BPM_OUT_HEADS || q.bias_m || q.resid ||
                q.gate || q.colsum || q.colsum_a || q.drop_p != 0.f || (q.flags & (BPM_GEMM_ATOMIC | BPM_GEMM_ACCUM)) ||
                (q.N & 3) || (q.ldc & 3) || (al & 15) || q.batch_stride_a < 0 || q.batch_stride_b < 0 || q.batch_stride_c < 0 ||
                ((q.batch_stride_a * sz) & 15) || ((q.batch_stride_b * sz) & 15) || (q.batch_stride_c & 3))
                return BPM_ERR_ARG;
            p.splitk = q.batch;
            p.hB = q.batch_stride_a; p.hH = q.batch_stride_b; p.hT = q.batch_stride_c;
        }
        tile += p.tiles_m * p.tiles_n * p.splitk;
    }
    g.total_tiles = tile;
    hipStream_t s = (hipStream_t)stream;
    double flops = 0, bytes = 0;
    for (int i = 0; i < nprob; ++i) {
        const bpm_gemm_problem& q = probs[i];
        flops += 2.0 * q.M * (double)q.N * q.K * ((q.flags & BPM_GEMM_BATCHED) ? q.batch : 1);
        // algorithmic HBM bytes: each operand once, the output once, each side operand of the epilogue once
        const double mn = (double)q.M * q.N;
        bytes += ((double)q.M + q.N) * q.K * sz + mn * (q.out_kind == BPM_OUT_F32 ? 4 : sz);
        if (q.resid) bytes += mn * 4;
        if (q.gate) bytes += mn * sz;
        if (q.out_kind == BPM_OUT_F32 && (q.flags & BPM_GEMM_ACCUM)) bytes += mn * 4;
    }
    BpmProfScope prof((dma >= 0 ? BPM_K_GEMM_DMA_NT : BPM_K_GEMM_NT) + variant, s, flops, bytes);
    if (x3 && dma < 0) return BPM_ERR_ARG;
    if (dma >= 0) return launch_dma(dma, variant, g, s, x3);
    if (skinny) {
        dim3 grid(g.total_tiles), block(256);
        if (dtype == BPM_BF16) {
            if (variant == BPM_GEMM_NT) hipLaunchKernelGGL((gemm_skinny_kernel<bf16_t, true>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_skinny_kernel<bf16_t, false>), grid, block, 0, s, g);
        } else {
            if (variant == BPM_GEMM_NT) hipLaunchKernelGGL((gemm_skinny_kernel<float, true>), grid, block, 0, s, g);
            else hipLaunchKernelGGL((gemm_skinny_kernel<float, false>), grid, block, 0, s, g);
        }
        BPM_CHECK_LAUNCH();
        return 0;
    }
    return dtype == BPM_BF16 ? launch<bf16_t>(variant, fast, bm_tile == 64, g, s)
                             : launch<float>(variant, fast, bm_tile == 64, g, s);
}
