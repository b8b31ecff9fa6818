// Launch profiler shared by the kernel files: when a kernel kind is enabled,
// its launches are bracketed by HIP events on the launch stream and their
// algorithmic work (FLOPs or bytes) is tallied.  Off by default; zero cost
// beyond one branch per launch.
#pragma once
#include <hip/hip_runtime.h>

enum { BPM_K_GEMM_NT = 0, BPM_K_GEMM_NN, BPM_K_GEMM_TN, BPM_K_ATTN_FWD, BPM_K_ATTN_BWD_DQ, BPM_K_ATTN_BWD_DKV,
       BPM_K_RESERVED6, BPM_K_LN_FWD, BPM_K_LN_BWD, BPM_K_ROWS_CAST, BPM_K_EMBED, BPM_K_GMU, BPM_K_PACK,
       BPM_K_GEMM_DMA_NT, BPM_K_GEMM_DMA_NN, BPM_K_GEMM_DMA_TN,      // the same products when the LDS-DMA kernel (gemm_dma.h) takes the launch
       BPM_K_COUNT };

extern unsigned g_bpm_prof_mask;
void bpm_prof_open(int kind, hipStream_t s, double work, double bytes);
void bpm_prof_close(int kind, hipStream_t s);

struct BpmProfScope {
    int kind; hipStream_t s; bool on;
    BpmProfScope(int k, hipStream_t st, double work, double bytes = 0.0) : kind(k), s(st), on((g_bpm_prof_mask >> k) & 1u) {
        if (on) bpm_prof_open(kind, s, work, bytes);
    }
    ~BpmProfScope() { if (on) bpm_prof_close(kind, s); }
};
