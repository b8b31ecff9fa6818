// Fused crossmodal / self attention for BPMulT on gfx950: forward, dQ, dK/dV.
//
// Semantics (reference multihead_attention.py:86-130, transformer.py:209-216):
//   S = Qs K^T  (Qs already carries the dh^-0.5 scale, applied by the Q-projection epilogue)
//   key j visible to query i  iff  j - i < mask_off   (mask_off = 1 + |S-T|, or "infinite" without attn_mask)
//   P = softmax_fp32(S) over visible keys;  Pd = dropout(P);  O = Pd V
// No key-padding mask: zero-padded key rows take part.
// The [T,S] score matrix never touches HBM: forward keeps a running (max, sum)
// per query and stores only LSE = max + ln(sum); backward recomputes P from
// Q, K and LSE.
//
// Layouts (HBM):  Q [B,H,T,dhp], K/V [B,H,S,dhp], dO [B,H,T,dhp]  -- CT,
// head-major, dhp = head_dim padded to 32 with zeros, written by the
// projection GEMM epilogues.  O, dQ, dK, dV are row-major [(t*B+b), ld]
// with column h*dh + c: the operand layout of the out-proj / in-proj GEMMs.
//
// MFMA orientation ("key on the rows" for forward and dQ, "key on the lane"
// for dK/dV; cdna_hip_programming.md Appendix B): the score tile is produced
// with the reduction index of the NEXT product on its register rows, so P / dS
// feed the second MFMA straight from the accumulator registers (pack_rows)
// and the other operand is read transposed from the LDS image (read_tr).
//   forward / dQ : S^T = K Qs^T   -> rows = keys, lane = query
//                  O^T  += V^T  Pd^T      dQ^T += K^T dS^T
//   dK/dV        : S   = Qs K^T   -> rows = queries, lane = key
//                  dV^T += dO^T Pd        dK^T += Qs^T dS
// One wave owns 16 queries (or 16 keys); a 256-thread workgroup owns 64.
//
// At head_dim 25 the kernels are VALU-bound (16 exponentials per 8 MFMAs), so
// the element loops are kept to a handful of instructions: visibility is one
// compare against a per-lane limit and is skipped altogether on tiles that a
// wave-uniform test proves fully visible; dropout hashing runs only when
// enabled; exponentials are raw v_exp_f32.
#include <type_traits>

#include "bpm_common.h"
#include "bpm_prof.h"
#include "../../include/bpmult_hip.h"

namespace {

constexpr int NTHREADS = 256;
constexpr int KT = 64;      // keys per tile (forward / dQ)
constexpr int BPM_ATTN_W128 = 2;    // waves per SIMD at head_dim 128 (unified 256-register budget: no AGPR copies)
constexpr int BPM_ATTN_WF = 5;      // waves per SIMD of the forward kernel at head_dim <= 32 (90 VGPRs; dQ / dKdV spill at 5)
constexpr int BPM_BASE_PRIO = 1;      // see gemm.hip
constexpr int BPM_ATTN_SETPRIO = 1;
constexpr int BPM_ATTN_QT = 64;
constexpr int BPM_ATTN_DKV_W32 = 4;
constexpr int BPM_ATTN_DKV_W64 = 3;
constexpr int BPM_ATTN_DQ_W32 = 5;
constexpr int BPM_ATTN_DQ_W128 = 3;
constexpr int BPM_ATTN_DQ_W64 = 4;
constexpr int QT = BPM_ATTN_QT;      // queries per tile (dK/dV): a multiple of 32

// Waves per SIMD each kernel is compiled for (register budget 512 / waves), per kernel (0 forward, 1 dQ, 2 dK/dV),
// compute type and padded head_dim: the largest occupancy at which the kernel does not spill inside its tile loop.
// Measured on MI355X (tools/attn_lab.py, six encoders, B*H = 96 heads each, T = S = 512, masked), bf16: dQ at head_dim 64
// 110 us at 4 waves (12 registers spilled) -> 98 at 3.  Both backward kernels now compute one k-step of their second
// product at a time (dQ: 112 registers at head_dim 64, 89 at 25, 166 at 128; dK/dV: 154 / 114 / 222), which fits one more
// wave per SIMD without spills.  The f32 (parity-mode) kernels need
// one wave fewer.
template <typename CT>
constexpr int attn_waves(int kernel, int dhp) {
    const bool bf = sizeof(CT) == 2;
    if (dhp <= 32) return kernel == 0 ? (bf ? BPM_ATTN_WF : 4) : (kernel == 2 ? (bf ? BPM_ATTN_DKV_W32 : 3) : (bf ? BPM_ATTN_DQ_W32 : 4));
    if (dhp <= 64) return kernel == 0 ? (bf ? 4 : 3) : kernel == 1 ? (bf ? BPM_ATTN_DQ_W64 : 3) : (bf ? BPM_ATTN_DKV_W64 : 2);
    return kernel == 0 ? (bf ? 3 : 2) : (kernel == 1 && bf ? BPM_ATTN_DQ_W128 : BPM_ATTN_W128);
}
constexpr float LOG2E = 1.4426950408889634f;

struct AProb {
    const char* Q; const char* K; const char* V;
    char* O; int ldo;
    float* lse;
    const char* dO;
    float* delta;
    char* dQ; int lddq;
    char* dK; int lddk;
    char* dV; int lddv;
    int B, H, T, S, dh;
    int mask_off;
    int qpos0, qstride;   // query row i sits at time qpos0 + i*qstride (mask rule only)
    float dq_scale;
    DropCfg drop;
    char* dS; char* Pd;       // dQ pass, optional: dS and the dropped probabilities, element (b, h, i, j) at b xs_b + h xs_h + i xs_q + j
    int xs_b, xs_h, xs_q;
    int blk0, nblk;     // block prefix / blocks per (b,h)
    int pair;           // workgroups take two blocks (b, nblk - 1 - b) instead of one
};
struct AGroup {
    const uint64_t* seedp;     // dropout seed read at execution time (BPM_SEED_INDIRECT), or nullptr
    int nprob;
    AProb p[BPM_MAX_GROUP];
};

template <typename CT, int DHP> struct Cfg {
    static constexpr int SZ = sizeof(CT);
    static constexpr int NKS = DHP / Tr<CT>::KSTEP;        // k-steps over head_dim
    static constexpr int ND = DHP / 16;                     // 16-wide head_dim tiles
    static constexpr int ROWB = DHP * SZ;                   // bytes per head row
    static constexpr int STRIDE = ROWB + Tr<CT>::TR_PAD_B;  // LDS image row stride
    static constexpr int CPR = ROWB / 16;                   // chunks per row
};

// cooperative copy of `rows` head rows (global row index row0.., bound nrows) into an LDS image
template <typename CT, int DHP, int ROWS>
BPM_DEV void load_rows(char* img, const char* src, int row0, int nrows, int tid) {
    typedef Cfg<CT, DHP> C;
    constexpr int N = ROWS * C::CPR;
#pragma unroll
    for (int c = tid; c < N; c += NTHREADS) {
        const int row = c / C::CPR, cc = c % C::CPR;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (row0 + row < nrows) v = *(const u32x4*)(src + (size_t)(row0 + row) * C::ROWB + cc * 16);
        *(u32x4*)(img + row * C::STRIDE + cc * 16) = v;
    }
}

// the same copy split in two: global -> registers (issued one tile ahead, behind the current tile's arithmetic)
// and registers -> LDS image.  Loads are unconditional from a clamped row (row 0 always exists) and zeroed by
// a select, so no branch sits between the load and its use.
template <typename CT, int DHP, int ROWS>
struct RowStage {
    typedef Cfg<CT, DHP> C;
    static constexpr int N = ROWS * C::CPR;
    static constexpr int PT = (N + NTHREADS - 1) / NTHREADS;
    u32x4 r[PT];
    // `src` is wave-uniform (a head's first row): buffer loads through a descriptor of nrows rows, so rows past the end
    // come back as zeros from the range check instead of a clamp + select per dword (16 v_cndmask per tile and wave)
    BPM_DEV void load(const char* src, int row0, int nrows, int tid) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrows * C::ROWB, 0x00020000);
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = tid + i * NTHREADS;
            const int row = c / C::CPR, cc = c % C::CPR;
            int voff = (row0 + row) * C::ROWB + cc * 16;
            if (N % NTHREADS != 0 && c >= N) voff = 0x7fffffff;
            r[i] = (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
        }
    }
    BPM_DEV void store(char* img, int tid) const {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int c = tid + i * NTHREADS;
            if (N % NTHREADS != 0 && c >= N) continue;
            *(u32x4*)(img + (c / C::CPR) * C::STRIDE + (c % C::CPR) * 16) = r[i];
        }
    }
};

// operand chunk straight from a head-major global row (zero beyond nrows)
template <typename CT, int DHP>
BPM_DEV typename Tr<CT>::frag load_frag(const char* src, int row, int nrows, int ks, int g) {
    typedef Cfg<CT, DHP> C;
    if (row >= nrows) return Tr<CT>::zero();
    return *(const typename Tr<CT>::frag*)(src + (size_t)row * C::ROWB + ks * 64 + g * 16);
}

BPM_DEV const AProb& pick(const AGroup& grp, int& bid) {
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.nprob; ++i)
        if (bid >= grp.p[i].blk0) pi = i;
    bid -= grp.p[pi].blk0;
    return grp.p[pi];
}

// ---------------------------------------------------------------------------
// Row-contiguous store of one wave's 16 x dh result tile.  The MFMA leaves lane (c = lane & 15, g = lane >> 4) with
// columns 16n + 4g + r of row c; written from there, every element is its own 2-byte request to L2 (64 lanes x 16
// instructions per wave): measured on MI355X at B*H = 576 heads of 64, T = S = 512, the forward launch spent 54 of its
// 105 us in those stores, the dQ pass 98 of 144 (plus the same pattern reading O for delta) and dK/dV 95 of 167 --
// 19 M requests per output tensor at ~300 G requests/s.  Instead the wave passes the tile through a private LDS block
// (rows padded by 16 bytes: conflict-free for the 8-byte writes and the 16-byte reads) and stores whole 16-byte chunks
// with consecutive lanes on consecutive chunks of a row (a 64-wide head row is one 128-byte line per 8 lanes).
// Head dims that are not whole 16-byte chunks (25 at hidden 300) go element-wise, lanes along the row.
// Row i of the tile lives at base + i * rstride (elements); rows >= nvalid are not written.
// ---------------------------------------------------------------------------
template <typename CT, int DHP>
BPM_DEV void store_rows16(char* blk, CT* base, size_t rstride, int nvalid, int dh, const f32x4 (&acc)[DHP / 16], float scale, int lane) {
    constexpr int SZ = sizeof(CT), RS = DHP * SZ + 16, EPC = 16 / SZ;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int n = 0; n < DHP / 16; ++n) {
        const f32x4 v = acc[n] * scale;
        char* dst = blk + c * RS + (16 * n + 4 * g) * SZ;
        if constexpr (SZ == 4) *(f32x4*)dst = v;
        else { bf16x4 o; o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3]; *(bf16x4*)dst = o; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // one wave: its LDS accesses execute in order; this also pins the compiler
    const bool wide = (dh % EPC) == 0 && (((uintptr_t)base | (uintptr_t)(rstride * SZ)) & 15) == 0;
    if (wide && dh == DHP) {                                // the usual case (head_dim 64 / 128): row / chunk by shifts, no
        constexpr int cpr = DHP / EPC, total = 16 * cpr;    // run-time division (that was ~60 VALU instructions per tile)
#pragma unroll
        for (int idx0 = 0; idx0 < total; idx0 += 64) {
            const int idx = idx0 + lane;
            const int row = idx / cpr, ch = idx % cpr;
            if (row < nvalid) *(u32x4*)(base + (size_t)row * rstride + ch * EPC) = *(const u32x4*)(blk + row * RS + ch * 16);
        }
    } else if (wide) {
        const int cpr = dh / EPC, total = 16 * cpr;
        for (int idx = lane; idx < total; idx += 64) {
            const int row = idx / cpr, ch = idx - row * cpr;
            if (row < nvalid) *(u32x4*)(base + (size_t)row * rstride + ch * EPC) = *(const u32x4*)(blk + row * RS + ch * 16);
        }
    } else {
        const int total = 16 * dh;
        for (int idx = lane; idx < total; idx += 64) {
            const int row = idx / dh, col = idx - row * dh;
            if (row < nvalid) base[(size_t)row * rstride + col] = *(const CT*)(blk + row * RS + col * SZ);
        }
    }
}
template <typename CT, int DHP> constexpr int store_rows16_bytes() { return 16 * (DHP * (int)sizeof(CT) + 16); }

BPM_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32: exp2(-inf) = 0, no denormal fix-up

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <typename CT, int DHP>
BPM_DEV void attn_fwd_block(const AProb& P, const DropCfg& drop, char* smem, const int bh, const int qb) {
    typedef Cfg<CT, DHP> C;
    typedef typename Tr<CT>::frag frag;
    char* kimg = smem;
    char* vimg = smem + KT * C::STRIDE;

    const int b = bh / P.H, h = bh % P.H;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                 // per-lane values are re-derived in each pass of the block-pair loop, not kept live across it
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int q0 = qb * 64 + wave * 16;               // wave-uniform first query
    const int q = q0 + c;                             // this lane's query
    const char* Qh = P.Q + (size_t)bh * P.T * C::ROWB;
    const char* Kh = P.K + (size_t)bh * P.S * C::ROWB;
    const char* Vh = P.V + (size_t)bh * P.S * C::ROWB;

    frag qf[C::NKS];
#pragma unroll
    for (int s = 0; s < C::NKS; ++s) qf[s] = load_frag<CT, DHP>(Qh, q, P.T, s, g);

    f32x4 o[C::ND];
#pragma unroll
    for (int n = 0; n < C::ND; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const int q_hi = min(P.T, qb * 64 + 64) - 1;
    const int jend = min(P.S, P.qpos0 + q_hi * P.qstride + P.mask_off);      // one past the last key any query of this block sees (mask_off < 2^30)
    const int ntile = (jend + KT - 1) / KT;
    const int lim = min(P.S, P.qpos0 + q * P.qstride + P.mask_off);          // this lane sees keys j < lim
    const int lim_min = min(P.S, P.qpos0 + q0 * P.qstride + P.mask_off);     // every lane of the wave sees keys j < lim_min
    const bool dropping = drop.thresh != 0;
    const uint32_t drow = ((uint32_t)bh * (uint32_t)P.T + (uint32_t)q) * (uint32_t)P.S;
    const bool pair_ok = (P.S & 3) == 0;               // row starts are multiples of 4: keys (4m .. 4m+3) are one hash quad

    RowStage<CT, DHP, KT> kst, vst;
    if (ntile > 0) { kst.load(Kh, 0, P.S, tid); vst.load(Vh, 0, P.S, tid); }
#pragma unroll 1
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        kst.store(kimg, tid);
        vst.store(vimg, tid);
        __syncthreads();
        if (kt + 1 < ntile) { kst.load(Kh, (kt + 1) * KT, P.S, tid); vst.load(Vh, (kt + 1) * KT, P.S, tid); }

        f32x4 st[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < C::NKS; ++s) a = Tr<CT>::mma(read_rowfrag<CT>(kimg, C::STRIDE, 16 * n, s, lane), qf[s], a);
            st[n] = a;
        }
        const int jb = kt * KT + 4 * g;                // key of element (n, r) is jb + 16n + r
        if (kt * KT + KT > lim_min) {                  // wave-uniform: only tiles that touch the mask edge pay for it
            const int rel = lim - jb;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[n][r] = (16 * n + r < rel) ? st[n][r] : -INFINITY;
        }
        // 3-input maxima (v_max3_f32): two chains of four instead of a tree of fifteen 2-input ones
        auto max3 = [](float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); };
        float mx = max3(max3(max3(max3(st[0][0], st[0][1], st[0][2]), st[0][3], st[1][0]), st[1][1], st[1][2]), st[1][3], st[2][0]);
        float my = max3(max3(max3(st[2][1], st[2][2], st[2][3]), st[3][0], st[3][1]), st[3][2], st[3][3]);
        mx = fmaxf(mx, my);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = fast_exp2((m_run - m_new) * LOG2E);
        const float mneg = -m_new * LOG2E;
        m_run = m_new;
        // 4-wide float arithmetic compiles to packed-f32 VALU (v_pk_fma_f32 / v_pk_add_f32: two lanes of work per
        // instruction); only the exponentials stay scalar
        f32x4 ps4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f32x4 e4 = st[n] * LOG2E + mneg;
            f32x4 p4;
#pragma unroll
            for (int r = 0; r < 4; ++r) p4[r] = fast_exp2(e4[r]);
            ps4 += p4;
            st[n] = p4;
        }
        const float psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
        // the probabilities leave each branch already packed as the PV operand (8 registers instead of 16 to merge: without
        // this the no-dropout path paid 16 v_mov per tile to meet the dropout path's register assignment)
        constexpr int NPF = KT / Tr<CT>::KSTEP;
        frag pf[NPF];
        if (dropping) {
            if (pair_ok) {                             // wave-uniform: r = 0..3 share one hash
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    float d0, d1, d2, d3;
                    bpm_drop_mult4(drop, drow + (uint32_t)(jb + 16 * n), d0, d1, d2, d3);
                    st[n] *= f32x4{d0, d1, d2, d3};
                }
            } else {
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[n][r] *= bpm_drop_mult(drop, drow + (uint32_t)(jb + 16 * n + r));
            }
#pragma unroll
            for (int ks = 0; ks < NPF; ++ks) pf[ks] = Tr<CT>::pack_rows(st, ks);
        } else {
#pragma unroll
            for (int ks = 0; ks < NPF; ++ks) pf[ks] = Tr<CT>::pack_rows(st, ks);
        }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int n = 0; n < C::ND; ++n) o[n] *= alpha;
        // O^T += V^T Pd^T : k = keys of this tile
        if (BPM_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO + 1);
#pragma unroll
        for (int ks = 0; ks < NPF; ++ks) {
#pragma unroll
            for (int n = 0; n < C::ND; ++n)
                o[n] = Tr<CT>::mma(Tr<CT>::read_tr(vimg, C::STRIDE, ks * Tr<CT>::KSTEP, 16 * n, lane, Tr<CT>::TR_CTILE), pf[ks], o[n]);
        }
        if (BPM_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    // l_run >= 1 (the row maximum contributes exp(0)): v_log_f32 / v_rcp_f32 (1 ulp) without the library's denormal / division fix-ups
    if (q < P.T && g == 0) P.lse[(size_t)bh * P.T + q] = m_run + __builtin_amdgcn_logf(l_run) * 0.6931471805599453f;
    static_assert(2 * KT * C::STRIDE >= 4 * store_rows16_bytes<CT, DHP>(), "one transpose block per wave in the K / V images");
    __syncthreads();                                   // every wave is done with the K / V images
    if (q0 < P.T)
        store_rows16<CT, DHP>(smem + wave * store_rows16_bytes<CT, DHP>(), (CT*)P.O + ((size_t)q0 * P.B + b) * P.ldo + h * P.dh,
                              (size_t)P.B * P.ldo, P.T - q0, P.dh, o, __builtin_amdgcn_rcpf(l_run), lane);
}

// Two 64-row blocks per workgroup, b and nblk - 1 - b: under the future mask block b has b + 1 (forward / dQ) or nblk - b
// (dK / dV) tiles, so every pair carries the same work, and the per-block lead-in (operand loads, first tile, result
// store: ~6 us of latency that four resident workgroups per CU cannot hide) is paid half as often.  Measured on MI355X,
// 576 heads of 64, T = S = 512, masked: see DESIGN.md section 5 (attention).
template <typename CT, int DHP>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(attn_waves<CT>(0, DHP), attn_waves<CT>(0, DHP)))) void attn_fwd_kernel(const AGroup grp) {
    typedef Cfg<CT, DHP> C;
    __shared__ __attribute__((aligned(16))) char smem[2 * KT * C::STRIDE];
    if (BPM_BASE_PRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const AProb& P = pick(grp, bid);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const int nb2 = P.pair ? (P.nblk + 1) >> 1 : P.nblk;
    const int bh = bid / nb2, first = P.pair ? bid % nb2 : P.nblk - 1 - bid % nb2, second = P.pair ? P.nblk - 1 - first : first;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass && second == first) break;
        attn_fwd_block<CT, DHP>(P, drop, smem, bh, pass ? second : first);
    }
}

// ---------------------------------------------------------------------------
// backward, dQ (and delta = rowsum(dO * O))
// ---------------------------------------------------------------------------
// XP: also write dS and the dropped probabilities (AProb::dS / Pd) -- the few-query-row groups of the engine (level 2 under
// dead-row elimination) continue from those instead of dK / dV: with two query rows dK and dV have rank two per head, and
// every key / value-side product factors through [rows, S] matrices (engine.EncoderGroupPlan, "low-rank key side").
// A separate instantiation: the stores would cost the T = S = 512 launches registers.
template <typename CT, int DHP, bool XP>
BPM_DEV void attn_bwd_dq_block(const AProb& P, const DropCfg& drop, char* smem, const int bh, const int qb) {
    typedef Cfg<CT, DHP> C;
    typedef typename Tr<CT>::frag frag;
    char* kimg = smem;
    char* vimg = smem + KT * C::STRIDE;

    const int b = bh / P.H, h = bh % P.H;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                 // per-lane values are re-derived in each pass of the block-pair loop, not kept live across it
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int q0 = qb * 64 + wave * 16;
    const int q = q0 + c;
    const char* Qh = P.Q + (size_t)bh * P.T * C::ROWB;
    const char* dOh = P.dO + (size_t)bh * P.T * C::ROWB;
    const char* Kh = P.K + (size_t)bh * P.S * C::ROWB;
    const char* Vh = P.V + (size_t)bh * P.S * C::ROWB;

    frag qf[C::NKS], dof[C::NKS];
    float delta = 0.f;
    const bool o_wide = (P.dh % Tr<CT>::EPC) == 0 && (((uintptr_t)P.O | (uintptr_t)((size_t)P.ldo * C::SZ)) & 15) == 0;
#pragma unroll
    for (int s = 0; s < C::NKS; ++s) {
        qf[s] = load_frag<CT, DHP>(Qh, q, P.T, s, g);
        dof[s] = load_frag<CT, DHP>(dOh, q, P.T, s, g);
        if (q < P.T) {
            const CT* orow = (const CT*)P.O + ((size_t)q * P.B + b) * P.ldo + h * P.dh;
            const int d0 = s * Tr<CT>::KSTEP + g * Tr<CT>::EPC;
            if (o_wide) {                              // whole 16-byte chunks of the O row (element loads: one L2 request each)
                if (d0 < P.dh) {
                    const frag of = *(const frag*)(orow + d0);
#pragma unroll
                    for (int j = 0; j < Tr<CT>::EPC; ++j) delta += Tr<CT>::to_f(dof[s][j]) * Tr<CT>::to_f(of[j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < Tr<CT>::EPC; ++j)
                    if (d0 + j < P.dh) delta += Tr<CT>::to_f(dof[s][j]) * Tr<CT>::to_f(orow[d0 + j]);
            }
        }
    }
    delta += __shfl_xor(delta, 16);
    delta += __shfl_xor(delta, 32);
    const float lse2 = (q < P.T) ? -P.lse[(size_t)bh * P.T + q] * LOG2E : 0.f;
    if (q < P.T && g == 0) P.delta[(size_t)bh * P.T + q] = delta;

    f32x4 dq[C::ND];
#pragma unroll
    for (int n = 0; n < C::ND; ++n) dq[n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int q_hi = min(P.T, qb * 64 + 64) - 1;
    const int jend = min(P.S, P.qpos0 + q_hi * P.qstride + P.mask_off);
    const int ntile = (jend + KT - 1) / KT;
    const int lim = min(P.S, P.qpos0 + q * P.qstride + P.mask_off);
    const int lim_min = min(P.S, P.qpos0 + q0 * P.qstride + P.mask_off);
    const bool dropping = drop.thresh != 0;
    const uint32_t drow = ((uint32_t)bh * (uint32_t)P.T + (uint32_t)q) * (uint32_t)P.S;
    const bool pair_ok = (P.S & 3) == 0;               // row starts are multiples of 4: keys (4m .. 4m+3) are one hash quad

    RowStage<CT, DHP, KT> kst, vst;
    if (ntile > 0) { kst.load(Kh, 0, P.S, tid); vst.load(Vh, 0, P.S, tid); }
#pragma unroll 1
    for (int kt = 0; kt < ntile; ++kt) {
        __syncthreads();
        kst.store(kimg, tid);
        vst.store(vimg, tid);
        __syncthreads();
        if (kt + 1 < ntile) { kst.load(Kh, (kt + 1) * KT, P.S, tid); vst.load(Vh, (kt + 1) * KT, P.S, tid); }
        const int jb = kt * KT + 4 * g;
        const bool edge = kt * KT + KT > lim_min;
        const int rel = lim - jb;
        // one k-step of the dQ product (32 keys in bf16, 16 in f32) at a time: its dS lives only until its MFMAs
        constexpr int NPK = Tr<CT>::KSTEP / 16;
#pragma unroll
        for (int ks = 0; ks < KT / Tr<CT>::KSTEP; ++ks) {
            f32x4 ds[NPK];
#pragma unroll
            for (int nn = 0; nn < NPK; ++nn) {
                const int n = ks * NPK + nn;
                f32x4 s_ = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < C::NKS; ++s) {
                    s_ = Tr<CT>::mma(read_rowfrag<CT>(kimg, C::STRIDE, 16 * n, s, lane), qf[s], s_);
                    dp = Tr<CT>::mma(read_rowfrag<CT>(vimg, C::STRIDE, 16 * n, s, lane), dof[s], dp);
                }
                float dm[4] = {1.f, 1.f, 1.f, 1.f};
                if (dropping) {
                    if (pair_ok) {
                        bpm_drop_mult4(drop, drow + (uint32_t)(jb + 16 * n), dm[0], dm[1], dm[2], dm[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dm[r] = bpm_drop_mult(drop, drow + (uint32_t)(jb + 16 * n + r));
                    }
                }
                f32x4 e4 = s_ * LOG2E + lse2;
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) e4[r] = (16 * n + r < rel) ? e4[r] : -INFINITY;   // exp2(-inf) = 0: masked before the exponential
                }
                f32x4 p4;
#pragma unroll
                for (int r = 0; r < 4; ++r) p4[r] = fast_exp2(e4[r]);
                const f32x4 dm4 = f32x4{dm[0], dm[1], dm[2], dm[3]};
                ds[nn] = p4 * (dp * dm4 - delta);
                if constexpr (XP) {
                    const int j0 = jb + 16 * n;                  // four consecutive keys (S % 4 == 0: all four or none exist)
                    if (P.dS && q < P.T && j0 < P.S) {
                        const size_t off = (size_t)b * P.xs_b + (size_t)h * P.xs_h + (size_t)q * P.xs_q + j0;
                        const f32x4 pd = p4 * dm4;
                        if constexpr (C::SZ == 4) {
                            *(f32x4*)((float*)P.dS + off) = ds[nn];
                            *(f32x4*)((float*)P.Pd + off) = pd;
                        } else {
                            bf16x4 o, w;
#pragma unroll
                            for (int r = 0; r < 4; ++r) { o[r] = (bf16_t)ds[nn][r]; w[r] = (bf16_t)pd[r]; }
                            *(bf16x4*)((bf16_t*)P.dS + off) = o;
                            *(bf16x4*)((bf16_t*)P.Pd + off) = w;
                        }
                    }
                }
            }
            // dQ^T += K^T dS^T
            if (BPM_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO + 1);
            const frag df = Tr<CT>::pack_rows(ds, 0);
#pragma unroll
            for (int n = 0; n < C::ND; ++n)
                dq[n] = Tr<CT>::mma(Tr<CT>::read_tr(kimg, C::STRIDE, ks * Tr<CT>::KSTEP, 16 * n, lane, Tr<CT>::TR_CTILE), df, dq[n]);
            if (BPM_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
        }
    }
    __syncthreads();                                   // every wave is done with the K / V images
    if (q0 < P.T)
        store_rows16<CT, DHP>(smem + wave * store_rows16_bytes<CT, DHP>(), (CT*)P.dQ + ((size_t)q0 * P.B + b) * P.lddq + h * P.dh,
                              (size_t)P.B * P.lddq, P.T - q0, P.dh, dq, P.dq_scale, lane);
}

// Two 64-row blocks per workgroup, b and nblk - 1 - b: under the future mask block b has b + 1 (forward / dQ) or nblk - b
// (dK / dV) tiles, so every pair carries the same work, and the per-block lead-in (operand loads, first tile, result
// store: ~6 us of latency that four resident workgroups per CU cannot hide) is paid half as often.  Measured on MI355X,
// 576 heads of 64, T = S = 512, masked: see DESIGN.md section 5 (attention).
template <typename CT, int DHP, bool XP>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(attn_waves<CT>(1, DHP), attn_waves<CT>(1, DHP)))) void attn_bwd_dq_kernel(const AGroup grp) {
    typedef Cfg<CT, DHP> C;
    __shared__ __attribute__((aligned(16))) char smem[2 * KT * C::STRIDE];
    if (BPM_BASE_PRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const AProb& P = pick(grp, bid);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const int nb2 = P.pair ? (P.nblk + 1) >> 1 : P.nblk;
    const int bh = bid / nb2, first = P.pair ? bid % nb2 : P.nblk - 1 - bid % nb2, second = P.pair ? P.nblk - 1 - first : first;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass && second == first) break;
        attn_bwd_dq_block<CT, DHP, XP>(P, drop, smem, bh, pass ? second : first);
    }
}

// ---------------------------------------------------------------------------
// backward, dK and dV
// ---------------------------------------------------------------------------
template <typename CT, int DHP>
BPM_DEV void attn_bwd_dkv_block(const AProb& P, const DropCfg& drop, char* smem, const int bh, const int kb) {
    typedef Cfg<CT, DHP> C;
    typedef typename Tr<CT>::frag frag;
    char* qimg = smem;
    char* doimg = smem + QT * C::STRIDE;
    float* s_lse = (float*)(smem + 2 * QT * C::STRIDE);      // -lse * log2(e)
    float* s_del = s_lse + QT;

    const int b = bh / P.H, h = bh % P.H;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));                 // per-lane values are re-derived in each pass of the block-pair loop, not kept live across it
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int j0 = kb * 64 + wave * 16;                // wave-uniform first key
    const int j = j0 + c;                              // this lane's key
    const char* Qh = P.Q + (size_t)bh * P.T * C::ROWB;
    const char* dOh = P.dO + (size_t)bh * P.T * C::ROWB;
    const char* Kh = P.K + (size_t)bh * P.S * C::ROWB;
    const char* Vh = P.V + (size_t)bh * P.S * C::ROWB;

    frag kf[C::NKS], vf[C::NKS];
#pragma unroll
    for (int s = 0; s < C::NKS; ++s) {
        kf[s] = load_frag<CT, DHP>(Kh, j, P.S, s, g);
        vf[s] = load_frag<CT, DHP>(Vh, j, P.S, s, g);
    }
    f32x4 dk[C::ND], dv[C::ND];
#pragma unroll
    for (int n = 0; n < C::ND; ++n) { dk[n] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // query i sees key j iff i >= ilo = j - mask_off + 1 (and i < T, j < S)
    // in ROW indices: row i sits at time qpos0 + i*qstride, so "time >= tmin" is "i >= ceil((tmin - qpos0) / qstride)"
    auto first_row = [&](int tmin) { const int a = tmin - P.qpos0; return a <= 0 ? 0 : (a + P.qstride - 1) / P.qstride; };
    const int ilo = (j < P.S) ? first_row(j - P.mask_off + 1) : (1 << 30);
    const int ilo_max = (j0 + 15 < P.S) ? first_row(j0 + 15 - P.mask_off + 1) : (1 << 30);   // wave-uniform: tiles at or above it need no test
    const int i_first = first_row(kb * 64 - P.mask_off + 1);                                   // first query row that sees any key of the block
    const int qt_lo = i_first / QT;
    const int qt_hi = (P.T + QT - 1) / QT;
    const bool dropping = drop.thresh != 0;
    const bool pair_ok = (P.S & 3) == 0;               // row starts are multiples of 4: keys (4m .. 4m+3) are one hash quad

    RowStage<CT, DHP, QT> qst, dost;
    float n_lse = 0.f, n_del = 0.f;                    // threads < QT: next tile's -lse*log2(e) and delta
    const float* lse_h = P.lse + (size_t)bh * P.T;
    const float* del_h = P.delta + (size_t)bh * P.T;
    auto stage = [&](int qt) {
        qst.load(Qh, qt * QT, P.T, tid);
        dost.load(dOh, qt * QT, P.T, tid);
        const int i = qt * QT + (tid & (QT - 1));
        const bool ok = i < P.T;
        const float l = lse_h[ok ? i : 0], dl = del_h[ok ? i : 0];
        n_lse = ok ? -l * LOG2E : 0.f;
        n_del = ok ? dl : 0.f;
    };
    if (qt_lo < qt_hi) stage(qt_lo);
#pragma unroll 1
    for (int qt = qt_lo; qt < qt_hi; ++qt) {
        __syncthreads();
        qst.store(qimg, tid);
        dost.store(doimg, tid);
        if (tid < QT) { s_lse[tid] = n_lse; s_del[tid] = n_del; }
        __syncthreads();
        if (qt + 1 < qt_hi) stage(qt + 1);
        const bool edge = (qt * QT < ilo_max) || (qt * QT + QT > P.T);
        // one k-step of the dV / dK products (32 queries in bf16, 16 in f32) at a time: its scores, probabilities and
        // dS live only until its MFMAs (half the registers of doing the whole 64-query tile first)
        constexpr int UPK = Tr<CT>::KSTEP / 16;
#pragma unroll
        for (int ks = 0; ks < QT / Tr<CT>::KSTEP; ++ks) {
            f32x4 pd[UPK], ds[UPK];
#pragma unroll
            for (int uu = 0; uu < UPK; ++uu) {
                const int u = ks * UPK + uu;
                f32x4 s_ = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < C::NKS; ++s) {
                    s_ = Tr<CT>::mma(read_rowfrag<CT>(qimg, C::STRIDE, 16 * u, s, lane), kf[s], s_);
                    dp = Tr<CT>::mma(read_rowfrag<CT>(doimg, C::STRIDE, 16 * u, s, lane), vf[s], dp);
                }
                const f32x4 l4 = *(const f32x4*)(s_lse + 16 * u + 4 * g);
                const f32x4 d4 = *(const f32x4*)(s_del + 16 * u + 4 * g);
                const int ib = qt * QT + 16 * u + 4 * g;       // query of element r is ib + r
                f32x4 e4 = s_ * LOG2E + l4;
                if (edge) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) e4[r] = (ib + r >= ilo && ib + r < P.T) ? e4[r] : -INFINITY;
                }
                f32x4 p4, dm4 = f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) p4[r] = fast_exp2(e4[r]);
                if (dropping) {
                    if (pair_ok) {
                        // Element (query ib + r, key j) is 16-bit lane j & 3 of the hash quad (query, j >> 2) -- the quads the
                        // forward / dQ kernels draw once per four KEYS.  Here a lane holds four QUERIES of one key: four quads.
                        // The four lanes of a DPP quad hold keys 4m .. 4m + 3, so they need the same four quads: lane c hashes
                        // the one of query ib + (c & 3) and the others come over DPP (quad_perm broadcast): one hash
                        // (3 quarter-rate multiplies) per lane and step instead of four.
                        uint32_t w0, w1;
                        bpm_hash64((((uint32_t)bh * (uint32_t)P.T + (uint32_t)(ib + (c & 3))) * (uint32_t)P.S + (uint32_t)j) >> 2, drop.key, w0, w1);
                        const bool hi_word = (j & 2) != 0, hi_half = (j & 1) != 0;
                        auto from = [&](auto R) {
                            constexpr int r = decltype(R)::value;
                            const uint32_t a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w0, 0x55 * r, 0xf, 0xf, false);
                            const uint32_t a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w1, 0x55 * r, 0xf, 0xf, false);
                            const uint32_t w = hi_word ? a1 : a0;
                            const uint32_t bits = hi_half ? (w >> 16) : (w & 0xFFFFu);
                            dm4[r] = bits < drop.thresh ? 0.0f : drop.inv_keep;
                        };
                        from(std::integral_constant<int, 0>{}); from(std::integral_constant<int, 1>{});
                        from(std::integral_constant<int, 2>{}); from(std::integral_constant<int, 3>{});
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            dm4[r] = bpm_drop_mult(drop, ((uint32_t)bh * (uint32_t)P.T + (uint32_t)(ib + r)) * (uint32_t)P.S + (uint32_t)j);
                    }
                }
                pd[uu] = p4 * dm4;
                ds[uu] = p4 * (dp * dm4 - d4);
            }
            if (BPM_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO + 1);
            const frag pf = Tr<CT>::pack_rows(pd, 0);
            const frag df = Tr<CT>::pack_rows(ds, 0);
#pragma unroll
            for (int n = 0; n < C::ND; ++n) {
                dv[n] = Tr<CT>::mma(Tr<CT>::read_tr(doimg, C::STRIDE, ks * Tr<CT>::KSTEP, 16 * n, lane, Tr<CT>::TR_CTILE), pf, dv[n]);
                dk[n] = Tr<CT>::mma(Tr<CT>::read_tr(qimg, C::STRIDE, ks * Tr<CT>::KSTEP, 16 * n, lane, Tr<CT>::TR_CTILE), df, dk[n]);
            }
            if (BPM_ATTN_SETPRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
        }
    }
    static_assert(2 * QT * C::STRIDE >= 4 * store_rows16_bytes<CT, DHP>(), "one transpose block per wave in the Q / dO images");
    __syncthreads();                                   // every wave is done with the Q / dO images
    if (j0 < P.S) {
        char* blk = smem + wave * store_rows16_bytes<CT, DHP>();
        store_rows16<CT, DHP>(blk, (CT*)P.dK + ((size_t)j0 * P.B + b) * P.lddk + h * P.dh, (size_t)P.B * P.lddk, P.S - j0, P.dh, dk, 1.f, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the block is read out before dV overwrites it
        store_rows16<CT, DHP>(blk, (CT*)P.dV + ((size_t)j0 * P.B + b) * P.lddv + h * P.dh, (size_t)P.B * P.lddv, P.S - j0, P.dh, dv, 1.f, lane);
    }
}

// Two 64-row blocks per workgroup, b and nblk - 1 - b: under the future mask block b has b + 1 (forward / dQ) or nblk - b
// (dK / dV) tiles, so every pair carries the same work, and the per-block lead-in (operand loads, first tile, result
// store: ~6 us of latency that four resident workgroups per CU cannot hide) is paid half as often.  Measured on MI355X,
// 576 heads of 64, T = S = 512, masked: see DESIGN.md section 5 (attention).
template <typename CT, int DHP>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(attn_waves<CT>(2, DHP), attn_waves<CT>(2, DHP)))) void attn_bwd_dkv_kernel(const AGroup grp) {
    typedef Cfg<CT, DHP> C;
    __shared__ __attribute__((aligned(16))) char smem[2 * QT * C::STRIDE + 2 * QT * 4];
    if (BPM_BASE_PRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const AProb& P = pick(grp, bid);
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    const int nb2 = P.pair ? (P.nblk + 1) >> 1 : P.nblk;
    const int bh = bid / nb2, first = P.pair ? bid % nb2 : bid % nb2, second = P.pair ? P.nblk - 1 - first : first;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass && second == first) break;
        attn_bwd_dkv_block<CT, DHP>(P, drop, smem, bh, pass ? second : first);
    }
}

// tuning hook (tools/attn_lab.py): bit k = kernel k (0 forward, 1 dQ, 2 dK/dV) pairs blocks
// bit k: kernel k (forward, dQ, dK/dV) takes two 64-row blocks per workgroup.  A constant in the product library; a
// -DBPM_LAB build (tools/attn_lab.py, the block-pairing test) can switch it through bpm_debug_attn_pair.
#ifdef BPM_LAB
int g_attn_pair = 7;
#else
constexpr int g_attn_pair = 7;
#endif

int fill(AGroup& g, const bpm_attn_problem* probs, int nprob, int blocks_over_S, uint64_t seed, int* total, int kernel) {
    if (nprob < 1 || nprob > BPM_MAX_GROUP || !probs) return BPM_ERR_ARG;
    g.nprob = nprob;
    g.seedp = bpm_seed_ptr(seed);
    int blk = 0;
    for (int i = 0; i < nprob; ++i) {
        const bpm_attn_problem& q = probs[i];
        AProb& p = g.p[i];
        if (q.B < 1 || q.H < 1 || q.T < 1 || q.S < 1 || q.dh < 1 || q.dh > q.dhp) return BPM_ERR_ARG;
        if (q.T > (1 << 22) || q.S > (1 << 22)) return BPM_ERR_ARG;      // index arithmetic is 32-bit (rows * 512 B per head fits 31 bits)
        if (q.dhp != probs[0].dhp) return BPM_ERR_ARG;
        p.Q = (const char*)q.Q; p.K = (const char*)q.K; p.V = (const char*)q.V;
        p.O = (char*)q.O; p.ldo = q.ldo; p.lse = q.lse;
        p.dO = (const char*)q.dO; p.delta = q.delta;
        p.dQ = (char*)q.dQ; p.lddq = q.lddq; p.dK = (char*)q.dK; p.lddk = q.lddk; p.dV = (char*)q.dV; p.lddv = q.lddv;
        p.B = q.B; p.H = q.H; p.T = q.T; p.S = q.S; p.dh = q.dh;
        p.mask_off = (q.mask_off > 0 && q.mask_off < (1 << 29)) ? q.mask_off : (1 << 29);
        if (q.q_pos0 < 0 || q.q_stride < 0 || q.q_pos0 > (1 << 24) || q.q_stride > (1 << 24)) return BPM_ERR_ARG;
        p.qpos0 = q.q_pos0; p.qstride = q.q_stride > 0 ? q.q_stride : 1;
        if ((long)p.qpos0 + (long)(q.T - 1) * p.qstride > (1l << 28)) return BPM_ERR_ARG;
        p.dq_scale = q.dq_scale;
        p.dS = (char*)q.dS; p.Pd = (char*)q.Pd; p.xs_b = q.xs_b; p.xs_h = q.xs_h; p.xs_q = q.xs_q;
        if (q.dS || q.Pd) {       // both, whole 4-key groups, 4-element-aligned rows that do not run into each other
            if (!q.dS || !q.Pd || (q.S & 3) || ((q.xs_b | q.xs_h | q.xs_q) & 3) || q.xs_b < 0 || q.xs_h < 0 || q.xs_q < q.S) return BPM_ERR_ARG;
            if ((((uintptr_t)q.dS | (uintptr_t)q.Pd) & 15) != 0) return BPM_ERR_ALIGN;
        }
        p.drop = bpm_make_drop(q.drop_p, seed, q.drop_site);
        p.nblk = ((blocks_over_S ? q.S : q.T) + 63) / 64;
        p.blk0 = blk;
        p.pair = (g_attn_pair >> kernel) & 1;
        blk += (p.pair ? (p.nblk + 1) >> 1 : p.nblk) * q.B * q.H;
    }
    *total = blk;
    return 0;
}

// sum over problems of B*H*dh * (number of visible (query, key) pairs): the useful
// multiply-adds of ONE attention product (P(T,S) = sum_i min(S, i + mask_off))
double useful_pair_flops(const bpm_attn_problem* probs, int nprob) {
    double tot = 0;
    for (int i = 0; i < nprob; ++i) {
        const bpm_attn_problem& q = probs[i];
        double pairs = 0;
        if (q.mask_off <= 0) pairs = (double)q.T * q.S;
        else
            for (int t = 0; t < q.T; ++t) {
                const long tt = (long)q.q_pos0 + (long)t * (q.q_stride > 0 ? q.q_stride : 1);
                pairs += (double)(tt + q.mask_off < (long)q.S ? tt + q.mask_off : q.S);
            }
        tot += pairs * q.dh * q.B * q.H;
    }
    return tot;
}

// algorithmic HBM bytes of one launch: every tensor it must read or write once (which: 0 forward, 1 dQ + delta, 2 dK / dV)
double attn_bytes(const bpm_attn_problem* probs, int nprob, int sz, int which) {
    double tot = 0;
    for (int i = 0; i < nprob; ++i) {
        const bpm_attn_problem& q = probs[i];
        const double bh = (double)q.B * q.H, qe = bh * q.T * q.dh * sz, ke = bh * q.S * q.dh * sz, st = bh * q.T * 4;
        if (which == 0) tot += 2 * qe + 2 * ke + st;                 // Q, K, V in; O, lse out
        else if (which == 1) tot += 4 * qe + 2 * ke + 2 * st;        // Q, K, V, O, dO, lse in; dQ, delta out
        else tot += 2 * qe + 4 * ke + 2 * st;                        // Q, K, V, dO, lse, delta in; dK, dV out
    }
    return tot;
}

template <typename CT>
int dispatch(int which, int dhp, const AGroup& g, int total, hipStream_t s) {
    dim3 grid(total), block(NTHREADS);
#define BPM_ATTN_CASE(D)                                                                                        \
    case D:                                                                                                     \
        if (which == 0) hipLaunchKernelGGL((attn_fwd_kernel<CT, D>), grid, block, 0, s, g);                     \
        else if (which == 1) hipLaunchKernelGGL((attn_bwd_dq_kernel<CT, D, false>), grid, block, 0, s, g);      \
        else if (which == 3) hipLaunchKernelGGL((attn_bwd_dq_kernel<CT, D, true>), grid, block, 0, s, g);       \
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<CT, D>), grid, block, 0, s, g);                            \
        break;
    switch (dhp) {
        BPM_ATTN_CASE(32)
        BPM_ATTN_CASE(64)
        BPM_ATTN_CASE(128)
        default: return BPM_ERR_ARG;
    }
#undef BPM_ATTN_CASE
    BPM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

#ifdef BPM_LAB
extern "C" int bpm_debug_attn_pair(int mask) {
    if (mask < 0 || mask > 7) return BPM_ERR_ARG;
    g_attn_pair = mask;
    return 0;
}
#endif

extern "C" int bpm_attn_fwd(int dtype, const bpm_attn_problem* probs, int nprob, uint64_t seed, void* stream) {
    AGroup g;
    int total = 0;
    int rc = fill(g, probs, nprob, 0, seed, &total, 0);
    if (rc) return rc;
    for (int i = 0; i < nprob; ++i)
        if (!probs[i].Q || !probs[i].K || !probs[i].V || !probs[i].O || !probs[i].lse) return BPM_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    BpmProfScope prof(BPM_K_ATTN_FWD, s, 4.0 * useful_pair_flops(probs, nprob), attn_bytes(probs, nprob, dtype == BPM_BF16 ? 2 : 4, 0));
    return dtype == BPM_BF16 ? dispatch<bf16_t>(0, probs[0].dhp, g, total, s) : dispatch<float>(0, probs[0].dhp, g, total, s);
}

// dQ first (it also produces delta), then dK/dV on the same stream.
// parts: 1 = dQ (+ delta), 2 = dK/dV (reads the delta a dQ pass wrote), 3 = both in that order
static int attn_bwd_parts(int dtype, const bpm_attn_problem* probs, int nprob, uint64_t seed, void* stream, int parts) {
    AGroup g;
    int total = 0;
    int rc = fill(g, probs, nprob, 0, seed, &total, 1);
    if (rc) return rc;
    for (int i = 0; i < nprob; ++i) {
        const bpm_attn_problem& q = probs[i];
        if (!q.Q || !q.K || !q.V || !q.O || !q.lse || !q.dO || !q.delta) return BPM_ERR_ARG;
        if ((parts & 1) && !q.dQ) return BPM_ERR_ARG;
        if ((parts & 2) && (!q.dK || !q.dV)) return BPM_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    // algorithmic backward = dP, dQ (first kernel) + dV, dK (second): 4 * pairs * dh each; recomputing S is overhead
    const double w = 4.0 * useful_pair_flops(probs, nprob);
    if (parts & 1) {
        BpmProfScope prof(BPM_K_ATTN_BWD_DQ, s, w, attn_bytes(probs, nprob, dtype == BPM_BF16 ? 2 : 4, 1));
        bool xp = false;
        for (int i = 0; i < nprob; ++i) xp = xp || probs[i].dS != nullptr;
        const int which = xp ? 3 : 1;
        rc = dtype == BPM_BF16 ? dispatch<bf16_t>(which, probs[0].dhp, g, total, s) : dispatch<float>(which, probs[0].dhp, g, total, s);
        if (rc) return rc;
    }
    if (parts & 2) {
        rc = fill(g, probs, nprob, 1, seed, &total, 2);
        if (rc) return rc;
        BpmProfScope prof(BPM_K_ATTN_BWD_DKV, s, w, attn_bytes(probs, nprob, dtype == BPM_BF16 ? 2 : 4, 2));
        rc = dtype == BPM_BF16 ? dispatch<bf16_t>(2, probs[0].dhp, g, total, s) : dispatch<float>(2, probs[0].dhp, g, total, s);
    }
    return rc;
}

extern "C" int bpm_attn_bwd(int dtype, const bpm_attn_problem* probs, int nprob, uint64_t seed, void* stream) {
    return attn_bwd_parts(dtype, probs, nprob, seed, stream, 3);
}
extern "C" int bpm_attn_bwd_dq(int dtype, const bpm_attn_problem* probs, int nprob, uint64_t seed, void* stream) {
    return attn_bwd_parts(dtype, probs, nprob, seed, stream, 1);
}
extern "C" int bpm_attn_bwd_dkv(int dtype, const bpm_attn_problem* probs, int nprob, uint64_t seed, void* stream) {
    return attn_bwd_parts(dtype, probs, nprob, seed, stream, 2);
}
