/* bpmult_hip.h -- C ABI of libbpmult_hip.so, the MI355X (gfx950) implementation
 * of the BPMulT forward/backward hot path.
 *
 * The reference (Damorgal/Biprojection-Multimodal-Transformer) is pure Python:
 * its "FFI" for this path is the set of torch ops called by
 * bpmult/models/{mmtr,transformer,multihead_attention,position_embedding}.py.
 * Each entry point below replaces the ops named in its comment (file:line into
 * the reference); INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer owned by
 *    the caller (PyTorch allocates); nothing here allocates or frees;
 *  - every function returns 0 on success, BPM_ERR_* or a hipError_t otherwise
 *    (bpm_error_string() renders either); nothing is printed;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *    the call returns without synchronising (graph-capturable);
 *  - `dtype` selects the compute type CT of MFMA operands: BPM_F32 (exact f32
 *    MFMA 16x16x4 -- parity mode) or BPM_BF16 (MFMA 16x16x32, f32 accumulate);
 *  - row-major CT buffers that feed a GEMM have a leading dimension that is a
 *    multiple of 32 elements and ZERO padding columns; residual-stream tensors
 *    are fp32 [(t*B + b), d] exactly as torch lays out [T,B,d];
 *  - dropout masks are a pure function of (seed, site, element index)
 *    (counter hash), so backward regenerates them; drop_p = 0 disables;
 *  - every `seed` argument is either a 63-bit value or BPM_SEED_INDIRECT |
 *    the address of a device uint64 that the kernels read WHEN THEY RUN: a
 *    launch sequence captured once into a hipGraph then replays with a fresh
 *    seed per step (the host stores it before each replay).  Both forms draw
 *    identical masks for the same seed value.
 */
#ifndef BPMULT_HIP_H
#define BPMULT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPM_ABI_VERSION 3
#define BPM_SEED_INDIRECT (1ull << 63) /* seed = BPM_SEED_INDIRECT | (uintptr_t)device pointer to the uint64 seed */
#define BPM_MAX_GROUP 18 /* problems per grouped launch (6 encoders of a level x 3 projections) */
#define BPM_GEMM_MAX_GROUP 24 /* bpm_gemm_grouped alone: 6 encoders x (q, k, v, out) weight gradients in one launch */

enum { BPM_F32 = 0, BPM_BF16 = 1,
       BPM_BF16X3 = 2 /* bpm_gemm_grouped only: split-bf16 operands (bpm_split_rows), three bf16 MFMA products per fp32 product */ };
enum { BPM_ERR_ARG = -1, BPM_ERR_ALIGN = -2 };

int bpm_version(void);
const char* bpm_error_string(int code);

/* ------------------------------------------------------------------------
 * Grouped GEMM with fused epilogue.
 * Replaces: F.linear in MultiheadAttention._in_proj / out_proj
 * (multihead_attention.py:130,152-158), fc1/fc2 (transformer.py:186-190),
 * nn.Conv1d k=1 projections (mmtr.py:456-458,748-750), the GMU linears
 * (mmtr.py:189-195), the time-axis nn.Linear maps (mmtr.py:507-508) and the
 * corresponding autograd backward GEMMs (train.py:394).
 *   BPM_GEMM_NT: C[M,N] = A[M,K] . B[N,K]^T     (A, B k-contiguous)
 *   BPM_GEMM_NN: C[M,N] = A[M,K] . B[K,N]
 *   BPM_GEMM_TN: C[M,N] = A[K,M]^T . B[K,N]
 * v = ((acc + bias_n[n] + bias_m[m]) * alpha); ReLU; gate; dropout; + resid.
 * ---------------------------------------------------------------------- */
enum { BPM_GEMM_NT = 0, BPM_GEMM_NN = 1, BPM_GEMM_TN = 2 };
enum { BPM_OUT_F32 = 0, BPM_OUT_CT = 1, BPM_OUT_HEADS = 2 };
/* BPM_GEMM_KPAD_ZERO: the caller promises that every row of a k-contiguous operand (A of NT/NN, B of NT) is
 * readable up to its leading dimension and holds ZEROS in [K, ld) -- what every CT buffer written by this
 * library satisfies.  It lets the kernel use hardware-bounded buffer loads with no k-tail masking. */
/* BPM_GEMM_BACKGROUND: this product is off the caller's critical path (weight gradients, work put on a side
 * stream).  Forward / data-gradient products otherwise raise their waves' issue priority (s_setprio) so that, when
 * kernels of two streams share a CU, the critical-path kernel is served first. */
/* BPM_GEMM_A_OVERLAP / BPM_GEMM_B_OVERLAP: the rows of that operand OVERLAP in memory -- its leading dimension is
 * smaller than its row length (K of a k-contiguous operand, M / N of a k-strided one).  Row r is still the elements
 * [r*ld, r*ld + length): the sliding windows of a strided 1-D convolution over a channels-last signal are exactly such
 * a matrix (ld = stride*Cin, length = taps*Cin), so the convolution and its two gradients are products on the signal
 * itself and no window matrix is materialised (frontend.py; reference mmtr.py:93-108).  Needs BPM_GEMM_KPAD_ZERO
 * (hardware-bounded loads only: the operand must be readable up to (rows-1)*ld + length), and K % 64 == 0 when the
 * overlapping operand is k-contiguous (its k tail would otherwise read the next window's data instead of zeros). */
/* BPM_GEMM_CT_NARROW: a BPM_OUT_CT output writes its N columns only (by default the pad columns [N, ldc) of every row
 * receive zeros): for outputs whose rows are interleaved with other tensors' rows (ldc = several logical rows). */
enum { BPM_GEMM_ACCUM = 1, BPM_GEMM_RELU = 2, BPM_GEMM_ATOMIC = 4, BPM_GEMM_KPAD_ZERO = 8, BPM_GEMM_BACKGROUND = 16,
       BPM_GEMM_A_OVERLAP = 32, BPM_GEMM_B_OVERLAP = 64, BPM_GEMM_CT_NARROW = 128, BPM_GEMM_BATCHED = 256 };

typedef struct bpm_gemm_problem {
    const void* A;          /* CT */
    const void* B;          /* CT */
    void* C;                /* fp32 (BPM_OUT_F32) or CT */
    int M, N, K;
    int lda, ldb, ldc;      /* elements */
    const float* bias_n;    /* [N] or NULL */
    const float* bias_m;    /* [M] or NULL (time-axis Linear) */
    const float* resid;     /* fp32 [M, ldr] added after dropout, or NULL */
    int ldr;
    const void* gate;       /* CT [M, ldg]: v = gate > 0 ? v * gate_scale : 0 (ReLU+dropout backward), or NULL */
    int ldg;
    float gate_scale;
    float alpha;
    float drop_p;           /* dropout on element index m*N + n, keyed by (call seed, drop_site) */
    uint32_t drop_site;
    float* colsum;          /* [N] += column sums of v (before + resid), by atomics; or NULL (bias gradients) */
    int flags;              /* BPM_GEMM_* */
    int out_kind;           /* BPM_OUT_* */
    int splitk;             /* >1 needs BPM_GEMM_ATOMIC + BPM_OUT_F32 into a zeroed / accumulating buffer */
    /* BPM_OUT_HEADS: row m = t*B + b, column n = h*dh + c  ->  C[b][h][t][c] of [B,H,T,dhp] */
    int heads_B, heads_H, heads_T, heads_dh, heads_dhp;
    /* BPM_GEMM_TN only: colsum_a[m] += sum_k A[k, m] (the bias gradient that belongs to a weight gradient
     * dW = dY^T X: A = dY), taken from the operand tiles already in registers by one extra MFMA against a vector
     * of ones in the workgroups of the first N tile.  Needs splitk == 1.  NULL = off. */
    float* colsum_a;
    /* BPM_GEMM_BATCHED: `batch` products of this shape in one problem; operands / output of element i start
     * i * batch_stride_{a,b,c} elements behind the pointers above.  Plain stores (BPM_OUT_F32 / BPM_OUT_CT), no side
     * operands, N % 4 == 0; the 128 x 64 kernel only. */
    int batch, batch_stride_a, batch_stride_b, batch_stride_c;
} bpm_gemm_problem;

int bpm_gemm_grouped(int dtype, int variant, const bpm_gemm_problem* probs /* host */, int nprob,
                     uint64_t seed, void* stream);

/* ------------------------------------------------------------------------
 * Fused attention (never materialises the [T,S] scores).
 * Replaces: torch.bmm / += attn_mask / F.softmax(float) / F.dropout / torch.bmm
 * (multihead_attention.py:110-126), buffered_future_mask (transformer.py:209-216)
 * and their backward.  Q [B,H,T,dhp] (pre-scaled by dh^-0.5), K/V [B,H,S,dhp],
 * dO [B,H,T,dhp]: CT head-major, dhp in {32,64,128}.  O, dQ, dK, dV: CT
 * row-major [(t*B+b), ld], column h*dh + c.  lse, delta: fp32 [B,H,T].
 * mask_off: key j visible to query i iff j - i < mask_off (1 + |S-T| with
 * attn_mask; <= 0 means no mask).
 * ---------------------------------------------------------------------- */
typedef struct bpm_attn_problem {
    const void* Q; const void* K; const void* V;
    void* O; int ldo;
    float* lse;
    const void* dO;         /* backward only */
    float* delta;           /* backward scratch [B,H,T] */
    void* dQ; int lddq;     /* receives d(q before scaling) = dQs * dq_scale */
    void* dK; int lddk;
    void* dV; int lddv;
    int B, H, T, S, dh, dhp;
    int mask_off;
    float dq_scale;
    float drop_p;           /* on P, element index ((b*H+h)*T + i)*S + j, keyed by (call seed, drop_site) */
    uint32_t drop_site;
    int q_pos0, q_stride;   /* query row i is time step q_pos0 + i*q_stride for the mask rule (stride 0 = 1): a
                               gathered subset of query rows keeps its original visibility */
    /* bpm_attn_bwd_dq only, optional (both or neither; S % 4 == 0): dS = P o (drop o dP - delta), the gradient of the
     * scores, and Pd = drop o P, the probabilities after dropout, as CT elements at b*xs_b + h*xs_h + i*xs_q + j.  With a
     * handful of query rows dK = dS^T Q and dV = Pd^T dO have rank T per head and the caller may continue from these
     * [rows, S] matrices instead of bpm_attn_bwd_dkv (multihead_attention.py:110-130 backward). */
    void* dS; void* Pd;
    int xs_b, xs_h, xs_q;
} bpm_attn_problem;

int bpm_attn_fwd(int dtype, const bpm_attn_problem* probs /* host */, int nprob, uint64_t seed, void* stream);
int bpm_attn_bwd(int dtype, const bpm_attn_problem* probs /* host */, int nprob, uint64_t seed, void* stream);
/* The two halves of bpm_attn_bwd as separate launches: _dq writes dQ and delta = rowsum(dO * O); _dkv reads that
 * delta and writes dK, dV.  Only dQ is on the backward critical path (dK / dV feed weight gradients and the
 * key/value-source gradient), so the engine runs _dkv on its side stream. */
int bpm_attn_bwd_dq(int dtype, const bpm_attn_problem* probs /* host */, int nprob, uint64_t seed, void* stream);
int bpm_attn_bwd_dkv(int dtype, const bpm_attn_problem* probs /* host */, int nprob, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------
 * Row kernels.  All are grouped: `n` problems (<= BPM_MAX_GROUP) per launch,
 * problem arrays live in HOST memory and are copied into the kernel arguments.
 * ---------------------------------------------------------------------- */

/* Input staging.  Replaces x.transpose(1,2) / F.dropout on the text features /
 * .permute(2,0,1) around the Conv1d projections (mmtr.py:741-753): src fp32
 * [B,T,C] -> CT [(t*B+b), ld] (zero pad columns); backward: dsrc[b,t,c] =
 * drop_mult * g[(t*B+b), c]. */
typedef struct bpm_pack_problem {
    const float* src; void* dst;            /* forward */
    const float* g; int ldg; float* dsrc;   /* backward */
    int B, T, C, ld;
    float drop_p; uint32_t drop_site;       /* element index (b*T + t)*C + c */
} bpm_pack_problem;
int bpm_pack_rows_fwd(int dtype, const bpm_pack_problem* probs, int n, uint64_t seed, void* stream);
int bpm_pack_rows_bwd(const bpm_pack_problem* probs, int n, uint64_t seed, void* stream);

/* fp32 master weights -> CT shadows with zero-padded leading dimension; the
 * descriptor table lives in DEVICE memory (built once, reused every step).
 * Row r: dst[r*dst_ld + c] = c < cols ? src[r*src_ld + c] * (colscale ? colscale[c] : 1) : 0 for c < ld.
 * colscale (fp32 [cols], device) folds a LayerNorm gain into the projection that follows it: the key / value
 * side of a crossmodal layer normalises the SAME embedded source in every layer (transformer.py:167-172), so
 * the engine normalises it once without affine and uses W' = W * gamma, b' = W beta + b per layer. */
typedef struct bpm_pack_desc {
    const void* src;
    void* dst;
    int rows, cols, ld, src_ld, dst_ld;
    unsigned blk0;          /* first block of this tensor; a block covers 1024 (row, c<ld) elements */
    const float* colscale;
} bpm_pack_desc;
int bpm_pack_weights(int dtype, const bpm_pack_desc* table_dev, int ndesc, unsigned total_blocks, void* stream);

/* Folded bias of the above: out[n] = b[n] + sum_c W[n*ldw + c] * beta[c], n < rows.  Device-resident table;
 * one wave per output row, 4 rows per block (blk0 = first block of the entry). */
typedef struct bpm_fold_desc {
    const float* W; const float* beta; const float* b; float* out;
    int rows, cols, ldw;
    unsigned blk0;
} bpm_fold_desc;
int bpm_fold_bias(const bpm_fold_desc* table_dev, int ndesc, unsigned total_blocks, void* stream);

/* Backward of the folding.  The weight-gradient GEMM against the un-affined normalised source gives
 * dWf = dY^T xhat and the bias column sums give dbf; this turns them into the gradients of the real parameters:
 *   dW[n,c] += dWf[n,c]*gamma[c] + dbf[n]*beta[c];   dbias[n] += dbf[n];
 *   dgamma[c] += sum_n dWf[n,c]*W[n,c];               dbeta[c] += sum_n dbf[n]*W[n,c]      (atomics per block)
 * dWf is dense [rows, cols]; W / dW have row stride ldw.  16 rows per block. */
typedef struct bpm_unfold_desc {
    const float* dWf; const float* dbf; const float* W; const float* gamma; const float* beta;
    float* dW; float* dbias; float* dgamma; float* dbeta;
    int rows, cols, ldw;
    unsigned blk0;
} bpm_unfold_desc;
/* store_dw != 0: dW is WRITTEN (this launch is the first writer of those rows this step: no zero-fill, no read). */
int bpm_unfold_grads(const bpm_unfold_desc* table_dev, int ndesc, unsigned total_blocks, int store_dw, void* stream);

/* Zero a list of fp32 segments with one launch (device-resident table, built once).  A training step whose gradients
 * start from zero (zero_grad / `p.grad = None`, train.py:384-385,396-398) does not clear the whole flat gradient buffer:
 * the first weight-gradient GEMM of each large matrix stores instead of accumulating (no BPM_GEMM_ACCUM), and only
 * the small tensors that are summed from several launches (biases, LayerNorm affines, ...) are cleared by this.
 * blk0 = first block of the segment; a segment of n elements takes bpm_zero_segment_blocks(n) blocks. */
typedef struct bpm_zero_desc {
    float* p;
    unsigned n;
    unsigned blk0;
} bpm_zero_desc;
int bpm_zero_segment_blocks(unsigned n);
int bpm_zero_segments(const bpm_zero_desc* table_dev, int ndesc, unsigned total_blocks, void* stream);

/* Encoder prologue.  Replaces embed_scale * x + embed_positions(x[:,:,0]) and
 * F.dropout (transformer.py:66-79; position_embedding.py:8-27,62-76):
 * out = dropout(scale*x + table[pos]), pos = t+1 if x[t,b,0] != 0 else 0.
 * `table` is the fp32 sinusoid table [table_rows >= T+1, d] built on the host.
 * Backward: x = dy, out = dx, dx (+)= scale * drop_mult * dy. */
typedef struct bpm_embed_problem {
    const float* x; float* out;
    int T, B;
    int accumulate;                         /* backward only */
    float drop_p; uint32_t drop_site;       /* element index (t*B + b)*d + c */
    int pos0, pos_stride;                   /* row t is time step pos0 + t*pos_stride (stride 0 = 1): a gathered
                                               subset of rows keeps its original positions */
} bpm_embed_problem;
int bpm_embed_pos_fwd(const bpm_embed_problem* probs, int n, const float* table, int table_rows, int d,
                      float scale, uint64_t seed, void* stream);
int bpm_embed_pos_bwd(const bpm_embed_problem* probs, int n, int d, float scale, uint64_t seed, void* stream);

/* LayerNorm (nn.LayerNorm(d), eps inside sqrt; transformer.py:91,153,167-172,
 * 183-185,227-229).  x fp32 [R,d].  Forward writes CT [R, ldo] with zero pad
 * columns, or plain fp32 [R, ldo] when out_f32; saves mean / rstd [R].
 * Backward: dx = add + dLN(dy); dgamma/dbeta += by atomics (both or neither).
 * Optional fused hand-off to the next backward GEMM (what bpm_rows_cast would
 * do with a = dx): cast = CT [R, ldc] copy of dx * dropout_mult(r*d + c) with
 * zero pad columns, cast_colsum[d] += its column sums (the bias gradient of
 * the linear whose output this residual branch was; transformer.py:174-175,
 * 189-190). */
typedef struct bpm_ln_problem {
    const float* x; const float* gamma; const float* beta;
    void* out; int ldo; int out_f32;
    float* mean; float* rstd;
    int R;
    const float* dy; int ldy; const float* add; float* dx; float* dgamma; float* dbeta;   /* backward */
    void* cast; int ldc; float* cast_colsum; float drop_p; uint32_t drop_site;           /* backward, optional */
} bpm_ln_problem;
int bpm_ln_fwd(int dtype, const bpm_ln_problem* probs, int n, int d, float eps, void* stream);
int bpm_ln_bwd(int dtype, const bpm_ln_problem* probs, int n, int d, uint64_t seed, void* stream);
/* Same with a caller-provided workspace of at least bpm_ln_bwd_ws_bytes(n, d) bytes (16-byte aligned, private to the
 * stream, ZERO when first used -- the library leaves its ticket words zero again): dgamma / dbeta / cast_colsum are then
 * produced from per-block partial rows by the last block of each problem to finish (single owner per column, fixed
 * order: bitwise reproducible) instead of float atomics from every block into the same rows.  ws == NULL: as bpm_ln_bwd.
 * The last block adds with a plain read-modify-write, so for the duration of the launch the stream must OWN those rows:
 * no other stream may accumulate into the same dgamma / dbeta / cast_colsum words concurrently (problems of one launch
 * that share a row are detected and fall back to atomics).  A launch that faults can leave ticket words non-zero:
 * re-zero (or re-create) the workspace after any error reported for its stream. */
size_t bpm_ln_bwd_ws_bytes(int n, int d);
int bpm_ln_bwd_ws(int dtype, const bpm_ln_problem* probs, int n, int d, uint64_t seed, void* ws, size_t ws_bytes, void* stream);

/* y = (a [+ b]) * dropout_mult(r*C + c); a is fp32 or (a_is_ct) CT.  Outputs,
 * each optional: CT copy [R, ldd] (pad zeroed), fp32 copy, column sums (+= by
 * atomics).  Residual-dropout backward + bias gradients (transformer.py:174-175,
 * 189-190), level 1->2 residual adds (mmtr.py:799-800). */
typedef struct bpm_cast_problem {
    const void* a; int lda; int a_is_ct;
    const float* b; int ldb;
    void* dst_ct; int ldd;
    int ct_cols;            /* CT columns written per row, [C, ct_cols) zeroed; 0 = ldd */
    float* dst_f32; int ldf;
    float* colsum;
    int R, C;
    float drop_p; uint32_t drop_site;
} bpm_cast_problem;
int bpm_rows_cast(int dtype, const bpm_cast_problem* probs, int n, uint64_t seed, void* stream);

/* fp32 [R, C] (row stride ld) -> split bf16 [R, 2 ldp] for the BPM_BF16X3 products: columns [0, ldp) = hi = bf16(x),
 * columns [ldp, 2 ldp) = lo = bf16(x - hi), pad columns [C, ldp) of both planes zero (ldp % 4 == 0; the LDS-DMA GEMM
 * wants ldp % 128 == 0).  The parity-grade mode of the linear layers (multihead_attention.py:152-158,
 * transformer.py:186-190, F.linear everywhere on the path): x y ~ hi hi + hi lo + lo hi, f32 accumulation. */
typedef struct bpm_split_problem {
    const float* src; void* dst;
    int R, C, ld, ldp;
} bpm_split_problem;
int bpm_split_rows(const bpm_split_problem* probs, int n, void* stream);

/* Fusion-GMU gating (GatedMultimodalLayerFeatures.forward, mmtr.py:189-195):
 * out = z*tanh(a1)*x1 + (1-z)*tanh(a2)*x2, z = sigmoid(ag); a1,a2,ag fp32 [R,d]
 * are the three bias-free linears (GEMM outputs).  Backward writes da1,da2,dag
 * as CT [R, ldg] and the direct terms dx1 = dout*z*tanh(a1), dx2 likewise. */
typedef struct bpm_gmu_problem {
    const float* a1; const float* a2; const float* ag; const float* x1; const float* x2;
    float* out;
    const float* dout; void* da1; void* da2; void* dag; int ldg; float* dx1; float* dx2;   /* backward */
    int R;
} bpm_gmu_problem;
/* out = sum of n_in fp32 tensors of `count` elements each (16-byte aligned pointers; out may alias an input).  The autograd sums where one tensor feeds several consumers (mmtr.py:779-847: a level-1 output is a Fusion-GMU
 * operand twice and a level-2 key / value source; a projected input feeds up to eight encoders), grouped per launch. */
#define BPM_ADDN_MAX 8
typedef struct bpm_addn_problem {
    float* out;
    const float* src[8];    /* BPM_ADDN_MAX */
    int n_in;
    size_t count;
} bpm_addn_problem;
int bpm_add_n(const bpm_addn_problem* probs, int n, void* stream);

/* Head-major q / dO [B,H,T,dhp] (the attention operands) -> block rows [(h*T + t)*B + b, ld]: columns [h*dh, (h+1)*dh)
 * hold the head's vector, every other column zero -- the per-head products of the engine's low-rank key side (groups
 * with a handful of query rows) then are plain grouped GEMMs over all heads.  dbias (optional, [H*dh], WRITTEN):
 * sum_{b,t} rowsum(Pd[(h*T+t)*B+b, 0:S]) * dO[b,h,t,:], the value-projection bias gradient (column sums of
 * dV = Pd^T dO; multihead_attention.py:100-104 backward). */
typedef struct bpm_expand_problem {
    const void* q; const void* dO;      /* CT [B,H,T,dhp] */
    void* qexp; void* dOexp;            /* CT [H*T*B, ld] */
    const void* Pd;                     /* CT [H*T*B, S] (bpm_attn_problem.Pd with xs_h = T*B*S, xs_q = B*S, xs_b = S) or NULL */
    float* dbias;                       /* or NULL */
    int B, H, T, S, dh, dhp, ld;
} bpm_expand_problem;
int bpm_expand_heads(int dtype, const bpm_expand_problem* probs /* host */, int n, void* stream);

int bpm_gmu2_fwd(const bpm_gmu_problem* probs, int n, int d, void* stream);
int bpm_gmu2_bwd(int dtype, const bpm_gmu_problem* probs, int n, int d, void* stream);

/* Front-end: AudioEncoder of the 4-modal model (mmtr.py:93-108: Conv1d(96,96,k=128,stride 2) x 2 + AdaptiveAvgPool1d(200)).
 * A strided convolution over a channels-last signal xc[(b,pos), ci] is bpm_gemm_grouped on the signal itself, its rows
 * read overlapping (BPM_GEMM_A_OVERLAP / _B_OVERLAP; no window matrix):
 *   y[(b,l), co] = sum_{k,ci} Wr[co, k*Cin + ci] xc[(b, stride*l + k), ci] + bias     NT, lda = stride*Cin, K = taps*Cin
 *   dWr = dy^T . windows (TN, the signal as overlapping B; + colsum_a = bias gradient);   dx: the same NT form over the
 *   zero-padded dy, one problem per output phase pos % stride.
 * bpm_signal_pack:   out[r, c] (CT, leading dim ld, r < total_rows): row r = b*rows_per_batch + front + l, l < L, takes
 *                    x[b*sb + c*sc + l*sl] (fp32, element strides); every other row is zeros.  C % 4 == 0.
 * bpm_signal_unpack: x[b*sb + c*sc + l*sl] = l < Lvalid ? src[(b*rows_per_batch + l)*ld + c] : 0 for l < L (fp32 both).
 * bpm_adaptive_pool1d_*: torch.nn.AdaptiveAvgPool1d over the position axis of a fp32 [(b,l), C] matrix -> [(b,i), C]. */
int bpm_signal_pack(int dtype, const float* x, void* out, int B, int C, int L, int64_t sb, int64_t sc, int64_t sl,
                    int front, int rows_per_batch, int64_t total_rows, int ld, void* stream);
int bpm_signal_unpack(const float* src, float* x, int B, int C, int L, int64_t sb, int64_t sc, int64_t sl,
                      int Lvalid, int rows_per_batch, int ld, void* stream);
int bpm_adaptive_pool1d_fwd(const float* y, float* out, int B, int C, int Lin, int Lout, void* stream);
int bpm_adaptive_pool1d_bwd(const float* dout, float* dy, int B, int C, int Lin, int Lout, void* stream);

/* The [B,d]-sized tail (all fp32, exact f32 VALU arithmetic from the fp32 master weights):
 *   x_i   = (top_i + mid_i)[0] + (top_i + mid_i)[N_i - 1]            level 1 -> 3 residual + token pick, mmtr.py:806-808
 *           (i = l, v, a in the order of mmtr.py:857); x_3 = extra (poster projection, 4-modal, mmtr.py:574)
 *   z_i   = sigmoid(G_i [x_0|..|x_{n-1}]),  t_i = tanh(W_i x_i),  h = sum_i z_i t_i      TextShifting{3,4}Layer, mmtr.py:197-247
 *   p1    = dropout(relu(proj1 h)),  y = proj2 p1 + h,  logits = out_layer y            mmtr.py:577-583 / 860-866
 * top_i / mid_i are [N_i, B, d]; Wh[i] = gmu.hidden{i+1}.weight [d,d], Wg[i] = gmu.x{i+1}_gate.weight [d, n d].
 * The forward keeps x, z, t, h, p1, y ([B, n d] or [B, d]) for the backward; z is the model's gate output. */
typedef struct bpm_tail_desc {
    int B, d, n, C;
    int N[3];
    const float* top[3]; const float* mid[3];
    const float* extra;
    const float* Wh[4]; const float* Wg[4];
    const float* W1; const float* b1; const float* W2; const float* b2; const float* Wo; const float* bo;
    float out_dropout; uint32_t drop_site;
    float* x; float* z; float* t; float* h; float* p1; float* y; float* logits;
} bpm_tail_desc;
/* Backward: dlogits [B,C] (and optionally dz) in; parameter gradients are ACCUMULATED (+=) into dWh / dWg / dW1 / db1 / dW2 / db2 / dWo / dbo;
 * rows 0 and N_i - 1 of dtop[i] / dmid[i] ([N_i,B,d], the other rows are never touched: keep them zero) and dextra
 * [B,d] (NULL for n = 3) are WRITTEN; dy, dp1, dh [B,d] and dzp, dtp, dx [B, n d] are scratch. */
typedef struct bpm_tail_grads {
    const float* dlogits;
    const float* dz;        /* gradient of the returned gates z [B, n d], or NULL */
    float* dWh[4]; float* dWg[4];
    float* dW1; float* db1; float* dW2; float* db2; float* dWo; float* dbo;
    float* dtop[3]; float* dmid[3]; float* dextra;
    float* dy; float* dp1; float* dh; float* dzp; float* dtp; float* dx;
} bpm_tail_grads;
int bpm_tail_fwd(const bpm_tail_desc* t, uint64_t seed, void* stream);
int bpm_tail_bwd(const bpm_tail_desc* t, const bpm_tail_grads* g, void* stream);

/* Fused Adam step over ONE flat fp32 buffer (replaces torch.optim.Adam's per-tensor loop of
 * train.py:123-125,396-398 for the trunk, whose parameters / gradients are views into flat buffers).
 * torch.optim.Adam semantics (no amsgrad, L2 weight decay folded into the gradient); `step` is the 1-based step
 * count for the bias corrections; grad_scale multiplies the gradient first (1/world after an all-reduce sum);
 * zero_grad != 0 clears the gradient in the same pass.  n % 4 == 0, 16-byte aligned pointers. */
int bpm_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float grad_scale, int zero_grad, void* stream);

/* The same step over the same flat buffers, table driven, WRITING THE CT WEIGHT SHADOWS as it stores the updated masters
 * (train.py:396-398 followed by the next forward's weight casts: one pass instead of two over the flat master).  The table
 * (device memory, built once) cuts the flat buffer into consecutive segments: off4 / n4 in units of 4 floats (16-byte
 * aligned), blk0 = first block (a segment takes bpm_adam_blocks(n4) blocks; entries sorted by blk0, covering the buffer).
 * dst != NULL: the segment starts with a whole [rows, cols] fp32 matrix (cols % 4 == 0) whose CT shadow is [rows, dst_ld];
 * shadow element (r, c) = CT(updated master (r, c)), pad columns are not touched.  dtype = the shadows' CT. */
typedef struct bpm_adam_seg {
    size_t off4;
    unsigned n4;
    unsigned blk0;
    void* dst;
    int rows, cols, dst_ld;
    int pad_;
} bpm_adam_seg;
int bpm_adam_blocks(size_t n4);
int bpm_adam_step_table(int dtype, const bpm_adam_seg* table_dev, int nseg, unsigned total_blocks, float* param, float* grad,
                        float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay,
                        int step, float grad_scale, int zero_grad, void* stream);

/* Engine plumbing (no reference counterpart): a non-blocking HIP stream at the device's lowest priority
 * (low_priority != 0) or at the default priority.  The host engine puts weight-gradient GEMMs and the
 * key/value-side chain there so that the dispatcher serves the critical-path stream first. */
int bpm_stream_create(int low_priority, void** out_stream);
int bpm_stream_priority_range(int* least, int* greatest);

/* ------------------------------------------------------------------------
 * Launch profiler (measurement aid for bench.py; no reference counterpart).
 * While bit `kind` of the mask is set, every launch of that kernel kind is
 * bracketed by two HIP events ON ITS LAUNCH STREAM and its algorithmic work is
 * tallied (GEMM: 2*M*N*K; attention: 2 FLOPs per visible (query,key) pair per
 * head-dim element per algorithmic product, recomputation not counted).
 * bpm_prof_collect waits for the recorded events, returns the summed
 * per-launch durations (ms), work and launch count of one kind, and clears them.
 * ---------------------------------------------------------------------- */
enum { BPM_PROF_GEMM_NT = 0, BPM_PROF_GEMM_NN = 1, BPM_PROF_GEMM_TN = 2, BPM_PROF_ATTN_FWD = 3,
       BPM_PROF_ATTN_BWD_DQ = 4, BPM_PROF_ATTN_BWD_DKV = 5,
       /* bpm_gemm_grouped launches that the LDS-DMA kernel takes (hidden >= 512 shapes) are tallied apart from the
        * 128 x 64 register-staged kernel's (kinds 0-2), so a kind is one kernel */
       BPM_PROF_GEMM_DMA_NT = 13, BPM_PROF_GEMM_DMA_NN = 14, BPM_PROF_GEMM_DMA_TN = 15 };
int bpm_prof_enable(unsigned kind_mask);
int bpm_prof_collect(int kind, double* total_ms, double* total_work, int* launches);
/* The same, plus the launches' ALGORITHMIC HBM bytes (GEMM: both operands once, the output once, every side operand of
 * the epilogue once -- residual, gate, the previous value for +=; attention: Q, K, V, O (+ dO, dQ, dK, dV) once). */
int bpm_prof_collect2(int kind, double* total_ms, double* total_work, double* total_bytes, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* BPMULT_HIP_H */
