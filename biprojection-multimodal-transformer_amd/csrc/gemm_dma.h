// LDS-DMA grouped GEMM for the hidden >= 512 shapes (bf16 operands, f32 accumulate).  Included by gemm.hip inside
// its anonymous namespace, after Prob / Group / the shared epilogue.
//
// Why a second kernel: at hidden 768 a step is ~18 TFLOP of GEMM whose k extents are 768 / 3072 / 4096.  The
// register-staged 128 x 64 kernel (gemm_tiled_kernel) was built for K = 300 (5-10 k iterations, latency bound,
// 5 workgroups per CU); at these sizes it spends its time staging: 64-byte half-line global loads, a ds_write pass
// per stage and 6 KB of LDS reads per 8 MFMAs.  This kernel
//   * computes a 128 x 128 ... 256 x 256 tile per workgroup, every wave a 64 x 64 or 128 x 64 block of MFMA 16x16x32
//     tiles (16 ds_read_b128 per 32 MFMAs / 24 per 64: at most half the LDS bytes per flop of the 128 x 64 kernel;
//     the vector-memory path of a CU moves 64 B per clock, so a 128 x 128 x 64 stage costs as many clocks to load as
//     its MFMAs take at peak and only the 256-wide tiles can run the matrix pipe above ~50 %),
//   * moves both operands global -> LDS with `buffer_load_dwordx4 ... lds` (no VGPRs, no ds_write pass), 64 k per
//     stage so that a k-contiguous row contributes one whole 128-byte line per stage,
//   * keeps NS stages in a ring: the loads of stage kt+NS-1 are issued right after the barrier that retires stage kt
//     and stay in flight across barriers (counted `s_waitcnt vmcnt(N)`, raw `s_barrier`: a `__syncthreads()` would
//     drain them), one barrier per k stage,
//   * shares the epilogue (bias, ReLU, gate, dropout, residual, column sums, f32 / CT / head-major stores) with the
//     tiled kernel: same swapped-operand MFMA, so a lane again owns 4 consecutive n of one output row.
//
// LDS images (one stage = X image then Y image, every image a multiple of 1 KiB = one wave-wide DMA piece):
//   k-contiguous operand ("row image"): [rows][128 B], 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7):
//     the 16 lanes a ds_read_b128 lane group serves (8 rows x chunk g, 8 rows x chunk g^1: MI355X_MICROARCH LDS table)
//     then fall on 16 distinct 16-byte slots of the 256-byte bank row.
//   k-strided operand ("col image"): 128-column sub-images [64 k][256 B], chunk ch of k-row r stored at
//     ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) (cdna_hip_programming.md T10, image (b)): conflict free for the
//     ds_read_b64_tr_b16 reads of the 16x16x32 operand (a half's two blocks are 8 k-rows apart).
// An LDS-DMA piece lands lane-linear (base + 16 * lane), so the swizzle is applied to the per-lane SOURCE address and
// again on the read -- never to the destination (rule 21 of the guide).
// Bounds: rows past M / k-rows past K are cut off by the buffer descriptor's range check (returns zeros; probed on
// gfx950: the check covers voffset + soffset, so the wave-uniform piece / stage offsets in soffset are checked too); column
// overhang of a k-strided operand only reaches output columns the epilogue masks.  k-contiguous operands must keep
// whole 128-byte stages inside their zero-padded rows (checked by the dispatcher).

constexpr int BPM_EPI_AHEAD = 1;       // 16-row epilogue steps whose side operands are in flight ahead of their use (8-wave configurations)
constexpr int BPM_DMA_EARLY = 1;       // see the k loop
#ifndef BPM_DMA_ABLATE
#define BPM_DMA_ABLATE 0      // lab builds only (tools/gemm_lab.py): 1 no MFMA, 2 no DMA in the loop, 4 no epilogue
#endif
#ifdef BPM_GEMM_TRACE
// diagnostic build only (tools/gemm_clock_probe.py): slot <- shader-clock counter, slot + 1 <- 100 MHz realtime counter
#define BPM_TRACE_CLK(slot) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { \
        g_trace[blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
        g_trace[blockIdx.x * 16 + (slot) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define BPM_TRACE_CLK(slot) do { } while (0)
#endif
constexpr int DK = 64;                     // k elements per stage
constexpr int DROW = 128;                  // bytes of k per row-image row

BPM_DEV int desc_bytes(int rows, int ld, int width, int sz, bool overlap = false) { return ((rows - 1) * ld + (overlap ? width : min(width, ld))) * sz; }

BPM_DEV int dma_row_off(int row, int c) { return row * DROW + ((c ^ ((row >> 1) & 7)) << 4); }
BPM_DEV int dma_col_off(int krow, int ch) { return krow * 256 + ((ch ^ (((krow & 3) << 2) | ((krow >> 2) & 3))) << 4); }

template <int N> BPM_DEV void dma_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------
// Wide epilogue.  The MFMA leaves a lane with 4 consecutive n of 16 different rows per instruction, so a direct store
// writes sixteen 64-byte (f32) / 32-byte (bf16) segments per wave instruction; at K = 768 that store pattern cost as
// much as the whole main loop (lab: NT out 80 us with, 37 us without its epilogue).  After the k loop the stage
// buffers are free: every wave passes its accumulators through a private 8 KB LDS block (32 rows x 64 f32, 16-byte
// chunk c of row r stored at c ^ (r & 15): conflict-free both ways) and comes back with lane l holding columns
// 4 (l & 15) .. + 3 of row l >> 4: each side-operand load and each store then covers four whole 256-byte (f32) /
// 128-byte (bf16) row segments.  Same arithmetic per element as epilogue_fast (bit-identical results); only used for
// problems that satisfy epi_fast_ok (the dispatcher checks it on the host).
// ---------------------------------------------------------------------------
BPM_DEV void flush_colsum_wide(const Prob& P, f32x4 cs, int nb, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = cs[q];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16 && nb + q < P.N) atomicAdd(P.colsum + nb + q, v);
    }
}

// side operands of rows mrow + 4 * i (i < NI), columns nb .. nb + 3: requested one 16-row step ahead of their use (the
// epilogue of a wave tile is a chain of TMW such steps, each a ~2 us round trip to memory when the loads sit right before
// the arithmetic: 10 steps at 160 rows per wave cost as much as the 12-stage main loop of a K = 768 product)
// OT: element type of CT outputs and of the gate operand (bf16; float for the bf16x3 products, whose activations are f32)
template <int NI, typename OT = bf16_t>
struct WideSide {
    typedef typename std::conditional<sizeof(OT) == 4, f32x4, bf16x4>::type gate_t;
    f32x4 addv[NI];
    gate_t gt[NI];
};

template <int NI, typename OT = bf16_t>
BPM_DEV void wide_load(const Prob& P, int mrow, int nb, WideSide<NI, OT>& s) {
    typedef typename WideSide<NI, OT>::gate_t gate_t;
    const uint32_t nbc = nb < P.N ? (uint32_t)nb : 0u;
    const bool accum = P.out_kind == BPM_OUT_F32 && (P.flags & BPM_GEMM_ACCUM);
    if (P.gate) {
#pragma unroll
        for (int i = 0; i < NI; ++i) s.gt[i] = *(const gate_t*)((const OT*)P.gate + (uint32_t)min(mrow + 4 * i, P.M - 1) * (uint32_t)P.ldg + nbc);
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) s.gt[i] = gate_t{};
    }
    if (P.resid) {
#pragma unroll
        for (int i = 0; i < NI; ++i) s.addv[i] = *(const f32x4*)(P.resid + (uint32_t)min(mrow + 4 * i, P.M - 1) * (uint32_t)P.ldr + nbc);
    } else if (accum) {
#pragma unroll
        for (int i = 0; i < NI; ++i) s.addv[i] = *(const f32x4*)((const float*)P.C + (uint32_t)min(mrow + 4 * i, P.M - 1) * (uint32_t)P.ldc + nbc);
    } else {
#pragma unroll
        for (int i = 0; i < NI; ++i) s.addv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// rows mrow + 4 * i (i < NI), columns nb .. nb + 3
template <int NI, typename OT = bf16_t>
BPM_DEV void wide_apply(const Prob& P, const DropCfg& drop, int mrow, int nb, const f32x4 (&acc)[NI], const f32x4 bias, const WideSide<NI, OT>& s, f32x4& csum) {
    const bool colok = nb < P.N;
    const bool f32out = P.out_kind == BPM_OUT_F32;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const EpiRow e = epi_row(P, mrow + 4 * i);
        const bool valid = e.ok && colok;
        f32x4 x = acc[i];
        if (valid) {
            x += bias;
            if (P.alpha != 1.f) x *= P.alpha;
            if (P.flags & BPM_GEMM_RELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = fmaxf(x[q], 0.f);
            }
            if (P.gate) {
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = (float)s.gt[i][q] > 0.f ? x[q] * P.gate_scale : 0.f;
            }
            if (drop.thresh != 0) {
                float d0, d1, d2, d3;
                bpm_drop_mult4(drop, e.didx + (uint32_t)nb, d0, d1, d2, d3);
                x[0] *= d0; x[1] *= d1; x[2] *= d2; x[3] *= d3;
            }
            csum += x;
            x += s.addv[i];
        }
        if (f32out) {
            if (valid) *(f32x4*)((float*)P.C + e.offc + nb) = x;
        } else if (P.out_kind == BPM_OUT_CT) {
            if (e.ok && nb < ((P.flags & BPM_GEMM_CT_NARROW) ? P.N : P.ldc)) {   // pad columns [N, ldc) receive zeros
                if (!valid) x = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (sizeof(OT) == 4) *(f32x4*)((float*)P.C + e.offc + nb) = x;
                else { bf16x4 o; o[0] = (bf16_t)x[0]; o[1] = (bf16_t)x[1]; o[2] = (bf16_t)x[2]; o[3] = (bf16_t)x[3]; *(bf16x4*)((bf16_t*)P.C + e.offc + nb) = o; }
            }
        } else if (valid) {                             // head-major
            uint32_t h = (uint32_t)nb / (uint32_t)P.hdh, c = (uint32_t)nb - h * (uint32_t)P.hdh;
            const uint32_t hstride = (uint32_t)(P.hT * P.hdhp);
            OT* base = (OT*)P.C + e.hrow;
            if ((P.hdh & 3) == 0) {                     // the 4 columns stay inside one head
                if constexpr (sizeof(OT) == 4) *(f32x4*)(base + h * hstride + c) = x;
                else { bf16x4 o; o[0] = (bf16_t)x[0]; o[1] = (bf16_t)x[1]; o[2] = (bf16_t)x[2]; o[3] = (bf16_t)x[3]; *(bf16x4*)(base + h * hstride + c) = o; }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    base[h * hstride + c] = (OT)x[q];
                    if (++c == (uint32_t)P.hdh) { c = 0; ++h; }
                }
            }
        }
    }
}

// One operand side of the workgroup tile: ROWS rows of the output (a multiple of 128), NW waves in the workgroup, DKT
// k elements per stage: 64 (128-byte rows: whole lines) or 32 (64-byte rows: the two-resident-workgroup configuration,
// whose three stages of a 256 x 128 tile are 72 KB).
template <bool KCONTIG, int ROWS, int NW, int DKT = DK>
struct DmaSide {
    static constexpr int ROWB = DKT * 2;                          // bytes of k per row-image row
    static constexpr int IMG_BYTES = ROWS * ROWB;                 // both layouts: ROWS * DKT k * 2 B
    static constexpr int PIECES = IMG_BYTES / 1024;
    static constexpr int PER_WAVE = PIECES / NW;
    static constexpr int RPP = 1024 / ROWB, SPR = ROWB / 16;      // row image: rows per 1 KB piece, 16-byte slots per row
    static constexpr int PPS = DKT / 4, SUB_BYTES = DKT * 256;    // col image: pieces per 128-column sub-image, its bytes
    static_assert(DKT == 64 || DKT == 32, "64 or 32 k per stage");
    static_assert(PIECES % NW == 0 && (KCONTIG || ROWS % 128 == 0), "pieces divide over the waves; k-strided images are 128-column sub-images");
    static_assert(NW == 4 || NW == 8 || NW == 16, "piece -> swizzle mapping assumes 4, 8 or 16 waves");
    static_assert(KCONTIG || NW <= PPS, "a wave's first piece lies in the first sub-image");

    // 16-byte chunk c of row r of a row image sits at chunk c ^ swz(r): conflict free for ds_read_b128's lane groups
    // (128-byte rows: see the header comment; 64-byte rows: the 128 x 64 kernel's swz4)
    static BPM_DEV int row_swz(int row) { return DKT == 64 ? ((row >> 1) & 7) : swz4(row); }
    static BPM_DEV int row_off(int row, int c) { return row * ROWB + ((c ^ row_swz(row)) << 4); }

    // Wave w moves pieces w, w + NW, ...  The per-lane byte offset of its piece 0 -- the other pieces differ by a
    // wave-uniform amount because the swizzle term repeats every 16 rows (row image) / 16 k-rows (col image).
    static BPM_DEV int voffset(int ld, int row0, int wave, int lane) {
        if (KCONTIG) {
            const int row = RPP * wave + lane / SPR, slot = lane % SPR;
            return (row0 + row) * ld * 2 + ((slot ^ row_swz(row)) << 4);
        }
        const int kr = 4 * wave + (lane >> 4), slot = lane & 15;      // piece = 4 k-rows of one 128-column sub-image
        const int x = ((kr & 3) << 2) | ((kr >> 2) & 3);
        return kr * ld * 2 + row0 * 2 + ((slot ^ x) << 4);
    }
    // wave-uniform byte offset of piece j relative to piece 0, and its LDS offset inside the image
    static BPM_DEV int piece_goff(int ld, int j) {
        if (KCONTIG) return j * RPP * NW * ld * 2;
        return ((NW * j) / PPS) * 256 + 4 * ((NW * j) % PPS) * ld * 2;     // sub-image (128 columns), then k-rows
    }
    static BPM_DEV int piece_lds(int wave, int j) {
        if (KCONTIG) return (wave + j * NW) * 1024;
        return ((NW * j) / PPS) * SUB_BYTES + (wave + ((NW * j) % PPS)) * 1024;
    }
    static BPM_DEV int stage_step(int ld) { return KCONTIG ? ROWB : DKT * ld * 2; }

    // DMA instruction j of this wave for one stage (soff: byte offset of the stage's k position).  A __device__
    // function: with the LDS-DMA builtin directly in the kernel's lambda the host pass silently drops the kernel stub.
    template <int J>
    static BPM_DEV void issue_one(__amdgpu_buffer_rsrc_t rsrc, char* img, int voff, int soff, int ld, int wave) {
        typedef __attribute__((address_space(3))) void* ldsp;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (ldsp)(img + piece_lds(wave, J)), 16, voff, soff + piece_goff(ld, J), 0, 0);
    }

    // MFMA operand of the 16 rows starting at r0 (multiple of 16), k-step ks (0 / 1) of the stage
    static BPM_DEV bf16x8 frag(const char* img, int r0, int ks, int lane) {
        const int r = lane & 15, g = lane >> 4;
        if (KCONTIG) return *(const bf16x8*)(img + row_off(r0 + r, 4 * ks + g));
        const int q = r >> 2, p = r & 3;
        const char* sub = img + (r0 >> 7) * SUB_BYTES;
        const int ch = ((r0 & 127) >> 3) + (p >> 1);
        const int k0 = 32 * ks + 8 * g + q;
        typedef bf16x4 __attribute__((address_space(3))) * lds4;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(sub + dma_col_off(k0, ch) + 8 * (p & 1)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(sub + dma_col_off(k0 + 4, ch) + 8 * (p & 1)));
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
        return f;
    }
};

// Part Q (0..3) of a stage's DMA instructions: the X pieces then the Y pieces of this wave, dealt over four quarters
// of the stage's MFMA work so that the issue cost of a DMA (tens of cycles each, MI355X_MICROARCH cycle constants)
// is spread between MFMA groups instead of heading the stage.
template <typename SX, typename SY, int Q, int F = 0>
BPM_DEV void dma_issue_part(__amdgpu_buffer_rsrc_t rsx, __amdgpu_buffer_rsrc_t rsy, char* bx, int vx, int vy, int sx, int sy,
                            int ldx, int ldy, int wave) {
    constexpr int LPS = SX::PER_WAVE + SY::PER_WAVE;
    if constexpr (F < LPS) {
        if constexpr ((F * 4) / LPS == Q) {
            if constexpr (F < SX::PER_WAVE) SX::template issue_one<F>(rsx, bx, vx, sx, ldx, wave);
            else SY::template issue_one<F - SX::PER_WAVE>(rsy, bx + SX::IMG_BYTES, vy, sy, ldy, wave);
        }
        dma_issue_part<SX, SY, Q, F + 1>(rsx, rsy, bx, vx, vy, sx, sy, ldx, ldy, wave);
    }
}

// WMD x WND waves, each a (16 TMW) x 64 block of the (16 TMW WMD) x (64 WND) workgroup tile; NS LDS stages.
// XS: some problem of the launch wants the column sums of X (bias gradient beside a weight gradient, TN only); without
// them the 8 / 4 extra accumulators are not carried (they spilled the 128 x 64 wave tile past its 256 registers).
// X3: the operands are split-bf16 images [rows, hi plane | lo plane] of fp32 matrices (bpm_split_rows; the leading
// dimension spans both planes, so a plane is ld / 2 elements = ld BYTES further) and the product is
// x y ~ hi hi + hi lo + lo hi: the k loop runs three segments of ceil(K / 64) stages over plane pairs (hi, hi), (hi, lo),
// (lo, hi) into the same accumulators; CT outputs and the gate operand are fp32.  Everything else is the same kernel.
// DKT = 32 (256 x 128 tiles, 8 waves, three 24 KB stages, <= 128 registers per wave): TWO workgroups fit a CU, so one's
// pipeline fill and epilogue -- memory phases during which its matrix pipes idle, 58 % of a K = 768 tile's life (DESIGN.md
// section 5) -- overlap the other's k loop, and a workgroup waiting at its stage barrier leaves the SIMDs to the other.
template <bool XK, bool YK, int WMD, int WND, int TMW, int NS, bool XS = false, bool X3 = false, int DKT = DK>
__global__ __launch_bounds__(64 * WMD * WND) __attribute__((amdgpu_waves_per_eu(DKT == 32 ? 4 : (WMD * WND + 3) / 4)))
void gemm_dma_kernel(const Group grp) {
    typedef typename std::conditional<X3, float, bf16_t>::type OT;
    static_assert(!XS || (!XK && !YK), "column sums of X belong to the weight-gradient product");
    constexpr int NW = WMD * WND, BMD = 16 * TMW * WMD, BND = 64 * WND, WROWS = 16 * TMW;
    typedef DmaSide<XK, BMD, NW, DKT> SX;
    typedef DmaSide<YK, BND, NW, DKT> SY;
    constexpr int KSPS = DKT / 32;                            // MFMA k-steps per stage
    constexpr int STAGE = SX::IMG_BYTES + SY::IMG_BYTES;
    constexpr int LPS = SX::PER_WAVE + SY::PER_WAVE;          // DMA instructions per wave and stage
    static_assert(NS == 2 || NS == 3, "2 or 3 stages");
    static_assert(DKT == 64 || !X3, "the split-operand products use 64-k stages");
    static_assert(TMW == 4 || TMW == 8 || TMW == 10, "wave tile 64 x 64, 128 x 64 or 160 x 64");
    __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WND, wn = wave % WND;

    BPM_TRACE_CLK(0);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const Prob& P = pick_problem(grp, bid);
    if (BPM_BASE_PRIO && XK && !(P.flags & BPM_GEMM_BACKGROUND)) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    const int m0 = (bid / P.tiles_n) * BMD, n0 = (bid % P.tiles_n) * BND;
    const int nk1 = (P.K + DKT - 1) / DKT;                // stages per plane pair
    const int nkt = X3 ? 3 * nk1 : nk1;

    // descriptor = exactly the bytes the operand owns: (rows - 1) leading dimensions plus the last row's width (whole k
    // stages of a k-contiguous row, whole 16-byte chunks of a k-strided one) -- an operand that is a COLUMN VIEW of a wider
    // buffer then never reads past the parent's last row (rows * ld from the view's first element would)
    // (X3: a split image is a whole allocation of rows x ld elements with zero pad columns in both planes: the range check
    // only has to cut off rows past M / N and k-rows past K)
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)P.X, 0, X3 ? (XK ? P.M : P.K) * P.ldx * 2 : desc_bytes(XK ? P.M : P.K, P.ldx, XK ? nkt * DKT : (P.M + 7) & ~7, 2, (P.flags & BPM_GEMM_A_OVERLAP) != 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)P.Y, 0, X3 ? (YK ? P.N : P.K) * P.ldy * 2 : desc_bytes(YK ? P.N : P.K, P.ldy, YK ? nkt * DKT : (P.N + 7) & ~7, 2, (P.flags & BPM_GEMM_B_OVERLAP) != 0), 0x00020000);
    const int vx = SX::voffset(P.ldx, m0, wave, lane), vy = SY::voffset(P.ldy, n0, wave, lane);
    const int stepx = SX::stage_step(P.ldx), stepy = SY::stage_step(P.ldy);
    const int ldx = P.ldx, ldy = P.ldy;

    // byte offsets of stage kt: its k position, and (X3) the planes of its segment -- (hi, hi), (hi, lo), (lo, hi)
    auto seg_of = [&](int kt) { return !X3 ? 0 : (kt >= 2 * nk1 ? 2 : (kt >= nk1 ? 1 : 0)); };
    auto part = [&](int kt, int buf, auto Q) {
        const int seg = seg_of(kt), kk = kt - seg * nk1;
        dma_issue_part<SX, SY, decltype(Q)::value>(rsx, rsy, smem + buf * STAGE, vx, vy, kk * stepx + (seg == 2 ? ldx : 0),
                                                   kk * stepy + (seg == 1 ? ldy : 0), ldx, ldy, wave);
    };
    auto stage = [&](int kt, int buf) {
        part(kt, buf, std::integral_constant<int, 0>{}); part(kt, buf, std::integral_constant<int, 1>{});
        part(kt, buf, std::integral_constant<int, 2>{}); part(kt, buf, std::integral_constant<int, 3>{});
    };

    f32x4 acc[4][TMW];                     // [n tile][m tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < TMW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient beside a weight gradient: column sums of X (see the tiled kernel), in the tiles of column block 0.  The
    // TMW m tiles of a wave row are dealt over its WND waves (XB each): one wave doing all of them is 25 % more MFMAs for
    // that wave, and with a barrier per stage the whole workgroup -- and, one tile per CU, the launch -- waits for it.
    constexpr int XB = (TMW + WND - 1) / WND;
    [[maybe_unused]] f32x4 xs[XB];
    [[maybe_unused]] bool do_xs = false;
    if constexpr (XS) {
        do_xs = P.colsum_x != nullptr && n0 == 0;
#pragma unroll
        for (int b = 0; b < XB; ++b) xs[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nkt) stage(s, s);

    int buf = 0;
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt == 1) BPM_TRACE_CLK(2);                               // first stage consumed: the pipeline is full
        // stage kt has landed once at most the younger stages' DMAs are outstanding (this wave's share), and for the
        // other waves' shares once every wave has passed the barrier behind that wait
        if (NS == 3 && kt + 1 < nkt) dma_wait<LPS>();
        else dma_wait<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // our reads of the buffer about to be refilled are done
        __builtin_amdgcn_s_barrier();
        const bool more = kt + NS - 1 < nkt;                         // wave-uniform
        const int nbuf = buf == 0 ? NS - 1 : buf - 1;                // the buffer stage kt-1 was read from
        const char* ix = smem + buf * STAGE;
        const char* iy = ix + SX::IMG_BYTES;
        auto step = [&](auto KS) {
            constexpr int ks = decltype(KS)::value;
            // (two-resident configuration with the bias column sums, 128 registers: the second pair of Y fragments is read
            // behind the first half of the MFMAs into the same registers -- with all four live it spilled 13 of them: 253 us
            // for the attention weight gradients, 148 so, the same as the 256 x 256 tile.  Without the column sums the late
            // reads cost: FFN weight gradients 279 -> 320 us, so those keep all four fragments up front)
            constexpr bool YLATE = DKT == 32 && XS;
            bf16x8 fx[TMW], fy[4];
#pragma unroll
            for (int b = 0; b < TMW; ++b) fx[b] = SX::frag(ix, wm * WROWS + 16 * b, ks, lane);
#pragma unroll
            for (int a = 0; a < (YLATE ? 2 : 4); ++a) fy[a] = SY::frag(iy, wn * 64 + 16 * a, ks, lane);
            if constexpr (XS) {
                if (do_xs && seg_of(kt) != 1) {            // (X3: segments 0 and 2 carry X's hi and lo planes; segment 1 repeats hi)
                    bf16x8 one;
#pragma unroll
                    for (int j = 0; j < 8; ++j) one[j] = (bf16_t)1.0f;
#pragma unroll
                    for (int b = 0; b < TMW; ++b)
                        if (b / XB == wn) xs[b % XB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(one, fx[b], xs[b % XB], 0, 0, 0);
                }
            }
            // 1: the 16-wave configuration and the weight-gradient product request the whole next stage right behind the
            // barrier (4 / 8 DMAs per wave) instead of a quarter of it between MFMA groups
            constexpr bool EARLY = BPM_DMA_EARLY != 0 && (NW == 16 || (!XK && !YK));
            if (more && !(BPM_DMA_ABLATE & 2)) {
                if constexpr (EARLY) {
                    if (ks == 0) stage(kt + NS - 1, nbuf);
                } else if constexpr (KSPS == 2) part(kt + NS - 1, nbuf, std::integral_constant<int, 2 * ks>{});
                else { part(kt + NS - 1, nbuf, std::integral_constant<int, 0>{}); part(kt + NS - 1, nbuf, std::integral_constant<int, 1>{}); }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < TMW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[a], fx[b], acc[a][b], 0, 0, 0);
            if (more && !(BPM_DMA_ABLATE & 2) && !EARLY) {
                if constexpr (KSPS == 2) part(kt + NS - 1, nbuf, std::integral_constant<int, 2 * ks + 1>{});
                else { part(kt + NS - 1, nbuf, std::integral_constant<int, 2>{}); part(kt + NS - 1, nbuf, std::integral_constant<int, 3>{}); }
            }
            if constexpr (YLATE) {
#pragma unroll
                for (int a = 2; a < 4; ++a) fy[a - 2] = SY::frag(iy, wn * 64 + 16 * a, ks, lane);
            }
#pragma unroll
            for (int a = 2; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < TMW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fy[YLATE ? a - 2 : a], fx[b], acc[a][b], 0, 0, 0);
        };
        if (!(BPM_DMA_ABLATE & 1)) {
            step(std::integral_constant<int, 0>{});
            if constexpr (KSPS == 2) step(std::integral_constant<int, 1>{});
        } else if (more) stage(kt + NS - 1, nbuf);
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    dma_wait<0>();                         // nothing of ours may still be writing LDS when the workgroup retires
    BPM_TRACE_CLK(4);

    const int r = lane & 15, g = lane >> 4;
    const int mw = m0 + wm * WROWS;
    if constexpr (XS) {
        if (do_xs && g == 0) {
#pragma unroll
            for (int b = 0; b < TMW; ++b) {
                const int m = mw + 16 * b + r;
                if (b / XB == wn && m < P.M) P.colsum_x[m] += xs[b % XB][0];
            }
        }
    }
    static_assert(NS * STAGE >= NW * 8192, "one 8 KB transpose block per wave");
    if (BPM_DMA_ABLATE & 4) {              // lab: no epilogue traffic (keep the accumulators alive)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < TMW; ++b) asm volatile("" ::"v"(acc[a][b]));
        return;
    }
    // every wave is done reading the last stage before anyone overwrites LDS with accumulators
    __builtin_amdgcn_s_barrier();
    char* blk = smem + wave * 8192;        // NS * STAGE >= NW * 8 KB for every configuration
    const int lr = lane >> 4, lc = lane & 15;
    const int nbw = n0 + wn * 64 + 4 * lc;
    f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (P.bias_n) bias = *(const f32x4*)(P.bias_n + (nbw < P.N ? nbw : 0));
    const DropCfg drop = bpm_resolve_drop(P.drop, grp.seedp);
    // side operands of 16-row step h live in sd[h % (AH + 1)], requested AH steps before their use -- where the register
    // budget allows: the 16-wave configuration (128 registers) spills with two sets and requests them right before their use
    constexpr int AH = (NW < 16 && DKT == 64) ? BPM_EPI_AHEAD : 0;
    WideSide<4, OT> sd[AH + 1];
#pragma unroll
    for (int h0 = 0; h0 < AH; ++h0)
        if (h0 < TMW) wide_load<4, OT>(P, mw + 16 * h0 + lr, nbw, sd[h0]);
    auto pass = [&](auto PB) {             // rows 32 PB .. 32 PB + 31 of the wave tile
        constexpr int pb = decltype(PB)::value;
        if constexpr (pb < TMW / 2) {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int row = 16 * bb + r;
                    *(f32x4*)(blk + row * 256 + (((4 * a + g) ^ (row & 15)) << 4)) = acc[a][2 * pb + bb];
                }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {           // 16 rows at a time: 4 row steps per lane in flight
                const int h = 2 * pb + hf;
                if (h + AH < TMW || AH == 0) wide_load<4, OT>(P, mw + 16 * (h + AH) + lr, nbw, sd[(h + AH) % (AH + 1)]);
                f32x4 v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 16 * hf + 4 * i + lr;
                    v[i] = *(const f32x4*)(blk + row * 256 + ((lc ^ (row & 15)) << 4));
                }
                wide_apply<4, OT>(P, drop, mw + 16 * h + lr, nbw, v, bias, sd[h % (AH + 1)], cs);
            }
        }
    };
    pass(std::integral_constant<int, 0>{}); pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{}); pass(std::integral_constant<int, 3>{});
    pass(std::integral_constant<int, 4>{});
    static_assert(TMW <= 10, "passes unrolled for up to 160 rows per wave");
    if (P.colsum) flush_colsum_wide(P, cs, nbw, lane);
#ifdef BPM_GEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BPM_TRACE_CLK(6);
#endif
}
