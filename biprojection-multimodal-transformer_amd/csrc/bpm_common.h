// Device-side helpers shared by every BPMulT gfx950 kernel.
//
// Compute type CT is either float (parity mode: exact f32 MFMA 16x16x4) or
// __bf16 (throughput mode: MFMA 16x16x32 with f32 accumulation).  Everything is
// written in units of a 16-byte "chunk" (8 bf16 / 4 f32) and a 64-byte "k-step"
// (32 bf16 / 16 f32), so both types share addressing.
//
// MFMA 16x16 lane maps (cdna_hip_programming.md section 3), lane l: r = l & 15, g = l >> 4
//   bf16 16x16x32: A[row r][k = 8g + j], B[k = 8g + j][col r], j = 0..7
//   f32  16x16x4 : A[row r][k = g],      B[k = g][col r]
//   C/D           : col = r, row = 4g + reg (reg = 0..3)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define BPM_DEV __device__ __forceinline__

// ---------------------------------------------------------------------------
// Counter-hash dropout.  keep(idx) is a pure function of (seed, site, idx), so
// forward and backward kernels with different tilings regenerate one mask.
// One 32-bit hash serves TWO consecutive elements (idx >> 1; the low / high 16 bits decide element
// idx & 3), a quarter of the integer work per element: P(drop) = round(p * 2^16) / 2^16.
// ---------------------------------------------------------------------------
struct DropCfg {
    uint32_t key;     // mixed (seed, site); 0 with thresh 0 when disabled
    uint32_t thresh;  // drop when the element's 16 hash bits < thresh
    float inv_keep;   // 1 / (1 - p)
};

// 64 hash bits of a counter: four 16-bit lanes, one per element of the quad (4q .. 4q+3).  Three 32-bit multiplies
// (quarter-rate VALU) per FOUR elements: the pair scheme this replaces spent three per two, which was 25 us of the 168 us
// fc1 launch at hidden 768 (ReLU + dropout epilogue over 75 M elements).
BPM_DEV void bpm_hash64(uint32_t q, uint32_t key, uint32_t& w0, uint32_t& w1) {
    uint32_t a = q + key;
    a ^= a >> 16;
    a *= 0x85EBCA6Bu;
    a ^= a >> 13;
    w0 = a * 0xC2B2AE35u;
    w0 ^= w0 >> 16;
    w1 = a * 0x27D4EB2Fu;
    w1 ^= w1 >> 15;
}
// fold (seed, site) into the 32-bit hash key.  Host side for a seed passed by value; device side when the seed is read
// from memory at execution time (BPM_SEED_INDIRECT: a captured hipGraph replays with a fresh seed per step)
__host__ __device__ static inline uint32_t bpm_host_drop_key(uint64_t seed, uint32_t site) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(site + 1u);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z ^ (z >> 32));
}
// `seed` argument of every bpm_* entry: a 63-bit value, or BPM_SEED_INDIRECT | address of a device uint64 the kernels
// read when they RUN.  Indirect: DropCfg.key carries the raw site and bpm_resolve_drop() finishes the key on the device
// with the same function, so both forms draw identical masks for the same seed value.
#define BPM_SEED_INDIRECT_BIT (1ull << 63)   // == BPM_SEED_INDIRECT of include/bpmult_hip.h
static inline const uint64_t* bpm_seed_ptr(uint64_t seed) {
    return (seed & BPM_SEED_INDIRECT_BIT) ? (const uint64_t*)(uintptr_t)(seed & ~BPM_SEED_INDIRECT_BIT) : nullptr;
}
static inline DropCfg bpm_make_drop(float p, uint64_t seed, uint32_t site) {
    DropCfg d;
    d.thresh = 0; d.key = 0; d.inv_keep = 1.f;
    if (p > 0.f) {
        d.thresh = (uint32_t)(p * 65536.0 + 0.5);
        d.key = (seed & BPM_SEED_INDIRECT_BIT) ? site : bpm_host_drop_key(seed, site);
        d.inv_keep = 1.f / (1.f - p);
    }
    return d;
}
BPM_DEV DropCfg bpm_resolve_drop(const DropCfg& c, const uint64_t* seedp) {
    DropCfg d = c;
    if (seedp != nullptr && c.thresh != 0) d.key = bpm_host_drop_key(*seedp, c.key);
    return d;
}
// multiplier applied to a kept element; 0 for a dropped one.  Element idx takes 16 bits of the hash of its quad idx >> 2:
// word (idx >> 1) & 1, low half for even idx, high half for odd.
BPM_DEV float bpm_drop_mult(const DropCfg& d, uint32_t idx) {
    if (d.thresh == 0) return 1.0f;
    uint32_t w0, w1;
    bpm_hash64(idx >> 2, d.key, w0, w1);
    const uint32_t w = (idx & 2u) ? w1 : w0;
    const uint32_t bits = (idx & 1u) ? (w >> 16) : (w & 0xFFFFu);
    return bits < d.thresh ? 0.0f : d.inv_keep;
}
// the quad (idx4 .. idx4 + 3) with one hash; idx4 must be a multiple of 4
BPM_DEV void bpm_drop_mult4(const DropCfg& d, uint32_t idx4, float& m0, float& m1, float& m2, float& m3) {
    uint32_t w0, w1;
    bpm_hash64(idx4 >> 2, d.key, w0, w1);
    m0 = (w0 & 0xFFFFu) < d.thresh ? 0.0f : d.inv_keep;
    m1 = (w0 >> 16) < d.thresh ? 0.0f : d.inv_keep;
    m2 = (w1 & 0xFFFFu) < d.thresh ? 0.0f : d.inv_keep;
    m3 = (w1 >> 16) < d.thresh ? 0.0f : d.inv_keep;
}

// ---------------------------------------------------------------------------
// per-type traits
// ---------------------------------------------------------------------------
template <typename CT> struct Tr;

template <> struct Tr<float> {
    typedef f32x4 frag;                 // one 16-byte chunk
    static constexpr int EPC = 4;       // elements per chunk
    static constexpr int KSTEP = 16;    // elements per 64-byte k-step
    static constexpr int CT_PER_KSTEP = 1;  // 16-row C sub-tiles that make one k-step
    static BPM_DEV float to_f(float v) { return v; }
    static BPM_DEV float from_f(float v) { return v; }
    static BPM_DEV frag zero() { return frag{0.f, 0.f, 0.f, 0.f}; }
    // acc += A(16 x KSTEP) * B(KSTEP x 16); element j of both frags is the same k
    static BPM_DEV f32x4 mma(frag a, frag b, f32x4 acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
        return acc;
    }
    // C sub-tiles (rows 4g+reg) -> operand chunk whose element j is row 4g + j of sub-tile ks
    static BPM_DEV frag pack_rows(const f32x4* ct, int ks) { return ct[ks]; }
    // Operand chunk read from a k-strided LDS image [krow][col] (col contiguous):
    // element j <- image[krow0 + 4g + j][col0 + r].  `second` is ignored for f32.
    static BPM_DEV frag read_tr(const char* img, int stride_b, int krow0, int col0, int lane, int /*second*/) {
        const int r = lane & 15, g = lane >> 4;
        const char* p = img + (size_t)(krow0 + 4 * g) * stride_b + (col0 + r) * 4;
        frag f;
        f[0] = *(const float*)(p);
        f[1] = *(const float*)(p + stride_b);
        f[2] = *(const float*)(p + 2 * stride_b);
        f[3] = *(const float*)(p + 3 * stride_b);
        return f;
    }
    static constexpr int TR_NATURAL = 0, TR_CTILE = 0;
    static constexpr int TR_PAD_B = 16;   // row pad (bytes) of a k-strided image
};

template <> struct Tr<bf16_t> {
    typedef bf16x8 frag;
    static constexpr int EPC = 8;
    static constexpr int KSTEP = 32;
    static constexpr int CT_PER_KSTEP = 2;
    static BPM_DEV float to_f(bf16_t v) { return (float)v; }
    static BPM_DEV bf16_t from_f(float v) { return (bf16_t)v; }
    static BPM_DEV frag zero() {
        frag f;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (bf16_t)0.f;
        return f;
    }
    static BPM_DEV f32x4 mma(frag a, frag b, f32x4 acc) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    // element j<4 is row 4g+j of sub-tile 2ks, element j>=4 is row 4g+j-4 of sub-tile 2ks+1
    static BPM_DEV frag pack_rows(const f32x4* ct, int ks) {
        frag f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[j] = (bf16_t)ct[2 * ks][j];
            f[4 + j] = (bf16_t)ct[2 * ks + 1][j];
        }
        return f;
    }
    // ds_read_b64_tr_b16 (T10): per 16-lane group, lane 4q+p supplies the
    // address of row q, columns 4p..4p+3; lane i receives column i of the 4 rows.
    //   second = 4  ("natural"): element j <- image[krow0 + 8g + j][col0 + r]
    //   second = 16 ("ctile")  : element j<4 <- image[krow0 + 4g + j][..], j>=4 <- image[krow0 + 16 + 4g + j-4][..]
    static BPM_DEV frag read_tr(const char* img, int stride_b, int krow0, int col0, int lane, int second) {
        const int i = lane & 15, g = lane >> 4;
        const int q = i >> 2, p = i & 3;
        const int row = krow0 + (second == 4 ? 8 * g : 4 * g) + q;
        const char* a0 = img + (size_t)row * stride_b + (col0 + 4 * p) * 2;
        const char* a1 = a0 + (size_t)second * stride_b;
        typedef bf16x4 __attribute__((address_space(3))) * lds4;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(a0));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(a1));
        frag f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
        return f;
    }
    static constexpr int TR_NATURAL = 4, TR_CTILE = 16;
    static constexpr int TR_PAD_B = 32;
};

// k-contiguous operand chunk from an LDS image [row][k] (row stride stride_b):
// lane (r, g) reads the 16 bytes at row (row0 + r), k-step byte offset ks*64 + 16g.
template <typename CT>
BPM_DEV typename Tr<CT>::frag read_rowfrag(const char* img, int stride_b, int row0, int ks, int lane) {
    const int r = lane & 15, g = lane >> 4;
    return *(const typename Tr<CT>::frag*)(img + (size_t)(row0 + r) * stride_b + ks * 64 + g * 16);
}

// Workgroups are dispatched round-robin over the 8 XCDs (workgroup i runs on XCD i % 8), each with its own L2.
// Map the hardware workgroup id to a logical id such that every XCD works through one CONTIGUOUS eighth of the
// logical ids: neighbouring tiles (which share operand panels) then share an L2 instead of being fetched 8 times.
BPM_DEV int xcd_remap(int hw, int total) {
    const int q = total >> 3, r = total & 7;
    const int x = hw & 7;
    return x * q + min(x, r) + (hw >> 3);
}

BPM_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// LDS byte offset of a generic pointer into __shared__ memory
BPM_DEV uint32_t lds_off(const void* p) { return (uint32_t)(uintptr_t)p; }

#define BPM_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)
