// Fused crossmodal-attention block for short sequences (gfx950) -- the north-star kernel of BASELINE.json:
//   Q = (xq Wq^T + bq) dh^-0.5,  K = xk Wk'^T + bk',  V = xv Wv'^T + bv'        multihead_attention.py:82-96
//   P = softmax_fp32(Q_h K_h^T + mask),  Pd = dropout(P),  O_h = Pd V_h         multihead_attention.py:110-126
//   out = resid + dropout(O Wo^T + bo)                                          multihead_attention.py:130, transformer.py:157-175
// in ONE launch for T, S <= 64 and head_dim 128 (hidden 768 / 6 heads, seq_len 50: the kernel point).  xq is the
// LayerNorm'ed query source, xk / xv the affine-free normalised key / value sources with the layer's LayerNorm gain and
// bias folded into Wk' / Wv' / bk' / bv' (engine.register_encoder_shadows): "LN-folded" projections.
//
// One 256-thread workgroup per (encoder, batch element): its 50 query rows and 50 key rows are ONE 64-row tile, so the
// four projections are 64 x 768 x 768 products whose weight tiles are the only real traffic (2 flop per weight byte
// and row: at 64 rows the 64 B/clk vector-memory path of a CU and the MFMA pipe are balanced).  Per head: three
// 64 x 128 x 768 products (operands global -> LDS by `buffer_load ... lds`, as gemm_dma.h) leave Q_h, K_h,
// V_h as bf16 in LDS (and in HBM, head-major, for the backward kernels); the 18 projections of the six heads are ONE
// stream of k stages through a 4-deep LDS ring, so three stages stay in flight across epilogues and attention phases
// (a 64 x 128 x 64 stage is 16 MFMAs per wave: the loop is paced by DMA latency, not arithmetic).  S^T = K Q^T, the
// masked fp32 softmax, dropout
// and O^T = V^T Pd^T then run from LDS exactly as attn_fwd_kernel does for one tile ("key on the rows": Pd feeds the
// second MFMA from the accumulator).  O goes to HBM (it is saved for backward anyway) and comes back through L2 as
// the A operand of the output projection, whose epilogue adds bias, dropout and the fp32 residual.
// The [T,S] scores, the per-head Q / K / V and O never make a round trip between launches: 5 launches -> 1.
#include "bpm_common.h"
#include "bpm_prof.h"
#include "../../include/bpmult_hip.h"

#ifndef BPM_BASE_PRIO
#define BPM_BASE_PRIO 1      // see gemm.hip
#endif

namespace {

constexpr int XT = 256;                    // threads
constexpr int XM = 64;                     // rows per workgroup (T, S <= 64)
constexpr int XDH = 128;                   // head_dim
constexpr int XK = 64;                     // k per stage
constexpr int A_IMG = XM * 128;            // 8 KB:  64 rows x 64 k bf16
constexpr int W_IMG = XDH * 128;           // 16 KB: 128 weight rows x 64 k
constexpr int STAGE = A_IMG + W_IMG;
constexpr int HEAD_IMG = XM * XDH * 2;     // 16 KB: [64][128] bf16
constexpr float LOG2E = 1.4426950408889634f;

struct XProb {
    const char* xq; const char* xk; const char* xv;
    const char* Wq; const char* Wk; const char* Wv; const char* Wo;
    const float* bq; const float* bk; const float* bv; const float* bo;
    const float* resid; float* out;
    char* qh; char* kh; char* vh; char* ao; float* lse;
    int B, H, T, S, d, ld, ldo, mask_off;
    float scale;
    DropCfg adrop, rdrop;
};
struct XGroup { int n; int blk0[BPM_MAX_GROUP + 1]; XProb p[BPM_MAX_GROUP]; };

BPM_DEV int row_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }          // stage images (gemm_dma.h)
BPM_DEV int head_off(int row, int ch) { return row * 256 + ((ch ^ (row & 15)) << 4); }              // Q_h / K_h: row reads
BPM_DEV int vimg_off(int key, int ch) { return key * 256 + ((ch ^ (((key & 3) << 2) | ((key >> 2) & 3))) << 4); }   // V_h: transposed reads

template <int N> BPM_DEV void xwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// this wave's DMA pieces of one k stage: rows of a k-contiguous bf16 matrix -> swizzled [rows][128 B] image
template <int ROWS>
BPM_DEV void dma_rows(__amdgpu_buffer_rsrc_t rsrc, char* img, int voff, int soff, int row_step_bytes, int wave) {
    typedef __attribute__((address_space(3))) void* ldsp;
    constexpr int PER_WAVE = ROWS * 128 / 1024 / 4;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (ldsp)(img + (wave + 4 * j) * 1024), 16, voff, soff + j * 32 * row_step_bytes, 0, 0);
}
// per-lane byte offset of this wave's piece 0: row 8 wave + lane / 8, source chunk = slot ^ swizzle(row)
BPM_DEV int dma_voff(int row_step_bytes, int wave, int lane) {
    const int row = 8 * wave + (lane >> 3), slot = lane & 7;
    return row * row_step_bytes + ((slot ^ ((row >> 1) & 7)) << 4);
}

struct GemmSrc { __amdgpu_buffer_rsrc_t rs; int voff; int step; };     // step: bytes between rows
constexpr int NSTG = 4;                    // LDS ring: three stages in flight behind the one being multiplied
constexpr int LPS = 6;                     // DMA instructions per wave and stage (2 for the 64 A rows, 4 for the 128 weight rows)

__global__ __launch_bounds__(XT) void xblock_fwd_kernel(const XGroup grp) {
    __shared__ __attribute__((aligned(1024))) char smem[NSTG * STAGE + 3 * HEAD_IMG];
    char* qimg = smem + NSTG * STAGE;
    char* kimg = qimg + HEAD_IMG;
    char* vimg = kimg + HEAD_IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.n; ++i)
        if (bid >= grp.blk0[i]) pi = i;
    const XProb& P = grp.p[pi];
    const int b = bid - grp.blk0[pi];
    if (BPM_BASE_PRIO) __builtin_amdgcn_s_setprio(BPM_BASE_PRIO);
    const int d = P.d, ld = P.ld, B = P.B, H = P.H, nkt = d / XK;
    const int astep = B * ld * 2;                          // bytes between rows t, t+1 of one batch element

    auto src_rows = [&](const char* X, int rows) {         // rows (t*B + b) of a [rows*B, ld] matrix
        GemmSrc s;
        s.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)b * ld * 2), 0, (rows * B - b) * ld * 2, 0x00020000);
        s.voff = dma_voff(astep, wave, lane);
        s.step = astep;
        return s;
    };
    const int wvoff = dma_voff(ld * 2, wave, lane);
    const GemmSrc aq = src_rows(P.xq, P.T), ak = src_rows(P.xk, P.S), av = src_rows(P.xv, P.S), ao = src_rows(P.ao, P.T);
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc((void*)P.Wq, 0, d * ld * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwk = __builtin_amdgcn_make_buffer_rsrc((void*)P.Wk, 0, d * ld * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwv = __builtin_amdgcn_make_buffer_rsrc((void*)P.Wv, 0, d * ld * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwo = __builtin_amdgcn_make_buffer_rsrc((void*)P.Wo, 0, d * ld * 2, 0x00020000);

    // The products of a phase form ONE stream of k stages (job j = product, 12 stages each): the DMAs of the next
    // product's first stages are in flight while this one's epilogue / the head's attention runs.
    // phase 0: jobs 3h + {0,1,2} = Q, K, V projection of head h;  phase 1: job c = output-projection column block c
    int p_job = 0, p_kt = 0, p_buf = 0, issued = 0;        // prefetch cursor
    int phase = 0, njobs = 3 * H;
    auto issue_next = [&]() {
        if (p_job >= njobs) return;
        char* base = smem + p_buf * STAGE;
        if (phase == 0) {
            const int h = p_job / 3, w = p_job - 3 * h;
            const int wsoff = h * XDH * ld * 2 + p_kt * 128;                  // weight rows of head h
            if (w == 0) { dma_rows<XM>(aq.rs, base, aq.voff, p_kt * 128, aq.step, wave); dma_rows<XDH>(rwq, base + A_IMG, wvoff, wsoff, ld * 2, wave); }
            else if (w == 1) { dma_rows<XM>(ak.rs, base, ak.voff, p_kt * 128, ak.step, wave); dma_rows<XDH>(rwk, base + A_IMG, wvoff, wsoff, ld * 2, wave); }
            else { dma_rows<XM>(av.rs, base, av.voff, p_kt * 128, av.step, wave); dma_rows<XDH>(rwv, base + A_IMG, wvoff, wsoff, ld * 2, wave); }
        } else {
            dma_rows<XM>(ao.rs, base, ao.voff, p_kt * 128, ao.step, wave);
            dma_rows<XDH>(rwo, base + A_IMG, wvoff, p_job * XDH * ld * 2 + p_kt * 128, ld * 2, wave);
        }
        ++issued;
        p_buf = p_buf + 1 == NSTG ? 0 : p_buf + 1;
        if (++p_kt == nkt) { p_kt = 0; ++p_job; }
    };
    int c_buf = 0, consumed = 0;
    f32x4 acc[4][2];
    // one product: acc (n tile a of 4, m tile b of 2: this wave's 32 x 64 block) = A[64 x d] W[128 x d]^T
    auto product = [&]() {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) acc[a][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kt = 0; kt < nkt; ++kt) {
            const int younger = issued - consumed - 1;     // stages issued behind the one needed now: 0 .. NSTG - 2
            if (younger >= 2) xwait<2 * LPS>();
            else if (younger == 1) xwait<LPS>();
            else xwait<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue_next();                                  // into the buffer the previous iteration read
            const char* ia = smem + c_buf * STAGE;
            const char* iw = ia + A_IMG;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[2], fw[4];
#pragma unroll
                for (int bt = 0; bt < 2; ++bt) fa[bt] = *(const bf16x8*)(ia + row_off(32 * wm + 16 * bt + r, 4 * ks + g));
#pragma unroll
                for (int a = 0; a < 4; ++a) fw[a] = *(const bf16x8*)(iw + row_off(64 * wn + 16 * a + r, 4 * ks + g));
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int bt = 0; bt < 2; ++bt) acc[a][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[a], fa[bt], acc[a][bt], 0, 0, 0);
            }
            ++consumed;
            c_buf = c_buf + 1 == NSTG ? 0 : c_buf + 1;
        }
    };

    // projection epilogue: + bias, * alpha -> bf16 into the LDS head image and head-major HBM
    auto put_head = [&](const float* bias, float alpha, char* img, bool vlayout, char* hbm, int rows, int h) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = 64 * wn + 16 * a + 4 * g;                       // head column of this lane's 4 values
            const f32x4 bv = *(const f32x4*)(bias + h * XDH + n);
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                const int m = 32 * wm + 16 * bt + r;
                const f32x4 v = (acc[a][bt] + bv) * alpha;
                bf16x4 o;
                o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
                const int ch = n >> 3, sub = (n & 7) * 2;
                *(bf16x4*)(img + (vlayout ? vimg_off(m, ch) : head_off(m, ch)) + sub) = o;
                if (m < rows) *(bf16x4*)(hbm + ((((size_t)b * H + h) * rows + m) * XDH + n) * 2) = o;
            }
        }
    };

    for (int i = 0; i < NSTG - 1; ++i) issue_next();
#pragma unroll 1
    for (int h = 0; h < H; ++h) {
        // (the head images of the previous head are free: every wave has passed a stage barrier since its attention)
        product();
        put_head(P.bq, P.scale, qimg, false, P.qh, P.T, h);
        product();
        put_head(P.bk, 1.f, kimg, false, P.kh, P.S, h);
        product();
        put_head(P.bv, 1.f, vimg, true, P.vh, P.S, h);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // head images complete (DMAs stay in flight)
        __builtin_amdgcn_s_barrier();

        // ---- attention of this head: wave = queries 16 wave .. 16 wave + 15, all 64 keys (attn_fwd_kernel, one tile)
        const int q = 16 * wave + r;
        bf16x8 qf[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) qf[s4] = *(const bf16x8*)(qimg + head_off(q, 4 * s4 + g));
        f32x4 st[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(kimg + head_off(16 * n + r, 4 * s4 + g)), qf[s4], a, 0, 0, 0);
            st[n] = a;
        }
        const int lim = min(P.S, q + P.mask_off);                         // this lane's query sees keys j < lim
        const int jb = 4 * g;                                             // key of element (n, e) is jb + 16 n + e
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) st[n][e] = (jb + 16 * n + e < lim) ? st[n][e] : -INFINITY;
        float mx = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])), fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(st[2][0], st[2][1]), fmaxf(st[2][2], st[2][3])), fmaxf(fmaxf(st[3][0], st[3][1]), fmaxf(st[3][2], st[3][3]))));
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mneg = -mx * LOG2E;
        float psum = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = __builtin_amdgcn_exp2f(st[n][e] * LOG2E + mneg);
                psum += pv;
                st[n][e] = pv;
            }
        psum += __shfl_xor(psum, 16);
        psum += __shfl_xor(psum, 32);
        const int bh = b * H + h;
        if (P.adrop.thresh != 0) {
            const uint32_t drow = ((uint32_t)bh * (uint32_t)P.T + (uint32_t)q) * (uint32_t)P.S;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int e = 0; e < 4; ++e) st[n][e] *= bpm_drop_mult(P.adrop, drow + (uint32_t)(jb + 16 * n + e));
        }
        f32x4 o[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                                  // 32 keys per step, "ctile" key order of the accumulators
            const bf16x8 pf = Tr<bf16_t>::pack_rows(st, ks);
            const int qq = r >> 2, pp = r & 3;
            typedef bf16x4 __attribute__((address_space(3))) * lds4;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int ch = 2 * n + (pp >> 1);
                const int k0 = 32 * ks + 4 * g + qq;
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(vimg + vimg_off(k0, ch) + 8 * (pp & 1)));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(vimg + vimg_off(k0 + 16, ch) + 8 * (pp & 1)));
                bf16x8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
                o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[n], 0, 0, 0);
            }
        }
        if (q < P.T) {
            const float inv = 1.f / psum;
            char* orow = P.ao + (((size_t)q * B + b) * P.ldo + h * XDH) * 2;
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)(o[n][e] * inv);
                *(bf16x4*)(orow + (16 * n + 4 * g) * 2) = ov;
            }
            if (g == 0) P.lse[(size_t)bh * P.T + q] = mx + logf(psum);
        }
    }

    // ---- output projection: the workgroup's own rows of O come back through L2 as the A operand
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // our O stores have left
    __syncthreads();                                                      // ... and everyone else's
    phase = 1; njobs = d / XDH; p_job = 0; p_kt = 0;
    for (int i = 0; i < NSTG - 1; ++i) issue_next();
#pragma unroll 1
    for (int c = 0; c < d / XDH; ++c) {
        product();
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = c * XDH + 64 * wn + 16 * a + 4 * g;
            const f32x4 bo = *(const f32x4*)(P.bo + n);
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) {
                const int t = 32 * wm + 16 * bt + r;
                if (t >= P.T) continue;
                const uint32_t m = (uint32_t)t * (uint32_t)B + (uint32_t)b;
                f32x4 v = acc[a][bt] + bo;
                if (P.rdrop.thresh != 0) {
                    float d0, d1, d2, d3;
                    bpm_drop_mult4(P.rdrop, m * (uint32_t)d + (uint32_t)n, d0, d1, d2, d3);
                    v[0] *= d0; v[1] *= d1; v[2] *= d2; v[3] *= d3;
                }
                v += *(const f32x4*)(P.resid + (size_t)m * d + n);
                *(f32x4*)(P.out + (size_t)m * d + n) = v;
            }
        }
    }
    xwait<0>();
}

}  // namespace

extern "C" int bpm_xblock_fwd(int dtype, const bpm_xblock_problem* q, int n, uint64_t seed, void* stream) {
    if (dtype != BPM_BF16 || !q || n < 1 || n > BPM_MAX_GROUP) return BPM_ERR_ARG;
    XGroup g;
    g.n = n; g.blk0[0] = 0;
    double flops = 0;
    for (int i = 0; i < n; ++i) {
        const bpm_xblock_problem& s = q[i];
        XProb& p = g.p[i];
        if (!s.xq || !s.xk || !s.xv || !s.Wq || !s.Wk || !s.Wv || !s.Wo || !s.bq || !s.bk || !s.bv || !s.bo || !s.resid || !s.out ||
            !s.qh || !s.kh || !s.vh || !s.ao || !s.lse) return BPM_ERR_ARG;
        if (s.B < 1 || s.T < 1 || s.S < 1 || s.T > XM || s.S > XM || s.H < 1 || s.d != s.H * XDH || s.ld != s.d || s.ldo != s.ld) return BPM_ERR_ARG;
        const uintptr_t al = (uintptr_t)s.xq | (uintptr_t)s.xk | (uintptr_t)s.xv | (uintptr_t)s.Wq | (uintptr_t)s.Wk | (uintptr_t)s.Wv |
                             (uintptr_t)s.Wo | (uintptr_t)s.bq | (uintptr_t)s.bk | (uintptr_t)s.bv | (uintptr_t)s.bo | (uintptr_t)s.resid |
                             (uintptr_t)s.out | (uintptr_t)s.qh | (uintptr_t)s.kh | (uintptr_t)s.vh | (uintptr_t)s.ao;
        if (al & 15) return BPM_ERR_ALIGN;
        if ((long)(s.T > s.S ? s.T : s.S) * s.B * s.ld * 2 >= (1l << 31)) return BPM_ERR_ARG;
        p.xq = (const char*)s.xq; p.xk = (const char*)s.xk; p.xv = (const char*)s.xv;
        p.Wq = (const char*)s.Wq; p.Wk = (const char*)s.Wk; p.Wv = (const char*)s.Wv; p.Wo = (const char*)s.Wo;
        p.bq = s.bq; p.bk = s.bk; p.bv = s.bv; p.bo = s.bo; p.resid = s.resid; p.out = s.out;
        p.qh = (char*)s.qh; p.kh = (char*)s.kh; p.vh = (char*)s.vh; p.ao = (char*)s.ao; p.lse = s.lse;
        p.B = s.B; p.H = s.H; p.T = s.T; p.S = s.S; p.d = s.d; p.ld = s.ld; p.ldo = s.ldo; p.mask_off = s.mask_off;
        p.scale = s.scale;
        p.adrop = bpm_make_drop(s.attn_drop, seed, s.attn_site);
        p.rdrop = bpm_make_drop(s.res_drop, seed, s.res_site);
        g.blk0[i + 1] = g.blk0[i] + s.B;
        // algorithmic flops: (4T + 4S) d^2 + 4 P(T,S) d per sample
        double pairs = 0;
        for (int t = 0; t < s.T; ++t) { int v = t + s.mask_off; pairs += v < s.S ? (v > 0 ? v : 0) : s.S; }
        flops += s.B * ((4.0 * s.T + 4.0 * s.S) * (double)s.d * s.d + 4.0 * pairs * s.d);
    }
    hipStream_t st = (hipStream_t)stream;
    BpmProfScope prof(BPM_K_XBLOCK, st, flops);
    hipLaunchKernelGGL(xblock_fwd_kernel, dim3(g.blk0[n]), dim3(XT), 0, st, g);
    BPM_CHECK_LAUNCH();
    return 0;
}
