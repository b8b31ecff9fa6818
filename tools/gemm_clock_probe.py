#!/usr/bin/env python3
"""In-kernel clock and phase times of the LDS-DMA GEMM (MI355X_MICROARCH, DVFS give-back item 6): a diagnostic build
stamps the shader-clock counter (s_memtime) and the 100 MHz realtime counter (s_memrealtime) per workgroup at kernel
start, after the first k stage, after the last one and after the epilogue's stores have drained.  Each case is launched
back to back for --seconds on random data before the stamps of its last launch are read.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBPM_GEMM_TRACE -o variants/lib_trace.so biprojection-multimodal-transformer_amd/csrc/*.hip
  BPMULT_LIB=variants/lib_trace.so python tools/gemm_clock_probe.py [--seconds 2] [--json out.json]

clock = d(s_memtime) / d(s_memrealtime) x 100 MHz over the k loop (median over workgroups); MFMA peak at that clock =
256 CUs x 4 SIMDs x 1024 flop/cycle x clock."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bpmult_amd  # noqa: F401,E402
from bpmult_amd import _lib, ops  # noqa: E402
from bpmult_amd.ops import BPM_BF16, F_ACCUM, F_KPAD, F_RELU, GEMM_NN, GEMM_NT, GEMM_TN, OUT_CT, OUT_F32  # noqa: E402

DEV, CT = "cuda", torch.bfloat16


def rc(r, c, scale=0.5, zero=False):
    return torch.zeros(r, c, device=DEV, dtype=CT) if zero else (torch.randn(r, c, device=DEV) * scale).to(CT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    lib = _lib.lib()
    if not hasattr(lib, "bpm_debug_trace"):
        sys.exit("needs the -DBPM_GEMM_TRACE build (see the docstring)")
    lib.bpm_debug_trace.argtypes = [C.c_void_p, C.c_int]
    d, R, G = 768, 4096, 6
    cases = []

    def add(name, variant, M, N, K, nprob, zero=False, **kw):
        probs, keep = [], []
        for _ in range(nprob):
            if variant == GEMM_NT:
                A_, B_ = rc(M, K, zero=zero), rc(N, K, K ** -0.5, zero)
            elif variant == GEMM_NN:
                A_, B_ = rc(M, K, zero=zero), rc(K, N, K ** -0.5, zero)
            else:
                A_, B_ = rc(K, M, zero=zero), rc(K, N, K ** -0.5, zero)
            ok = kw.get("out_kind", OUT_F32)
            C_ = torch.zeros(M, N, device=DEV, dtype=CT if ok == OUT_CT else torch.float32)
            ex = {}
            if kw.get("bias"):
                ex["bias_n"] = torch.randn(N, device=DEV)
            if kw.get("resid"):
                ex["resid"], ex["ldr"] = torch.randn(M, N, device=DEV), N
            keep += [A_, B_, C_] + [v for v in ex.values() if torch.is_tensor(v)]
            probs.append(ops.gemm_problem(A_, B_, C_, M, N, K, A_.shape[1], B_.shape[1], N, out_kind=ok, flags=kw.get("flags", 0) | F_KPAD,
                                          drop_p=kw.get("drop_p", 0.0), drop_site=3, **ex))
        cases.append((name, variant, ops.array(ops.GemmProblem, probs), keep, 2.0 * M * N * K * nprob))

    add("NT 4096^3 random", GEMM_NT, 4096, 4096, 4096, 1)
    add("NT 4096^3 zeros", GEMM_NT, 4096, 4096, 4096, 1, zero=True)
    add("NT 8192x8192x4096 random", GEMM_NT, 8192, 8192, 4096, 1, out_kind=OUT_CT)
    add("NT out (bias,resid,drop) x6", GEMM_NT, R, d, d, G, bias=True, resid=True, drop_p=0.1)
    add("NT fc1 (relu,drop,CT) x6", GEMM_NT, R, 4 * d, d, G, out_kind=OUT_CT, bias=True, flags=F_RELU, drop_p=0.1)
    add("NT fc2 (bias,resid,drop) x6", GEMM_NT, R, d, 4 * d, G, bias=True, resid=True, drop_p=0.1)
    add("NN dfc1 (f32) x6", GEMM_NN, R, d, 4 * d, G)
    add("TN wg ffn x12 (accum)", GEMM_TN, d, 4 * d, R, 2 * G, flags=F_ACCUM)
    out = []
    for name, variant, arr, _keep, flops in cases:
        fn = lambda: ops.gemm_grouped(BPM_BF16, variant, arr, 7)
        fn()
        torch.cuda.synchronize()
        t0, n = time.time(), 0
        while time.time() - t0 < a.seconds:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        buf = np.zeros((8192, 16), dtype=np.uint64)
        assert lib.bpm_debug_trace(buf.ctypes.data, 8192) == 0
        t = buf[buf[:, 1] > 0].astype(np.int64)
        # the last launch's workgroups: realtime stamps within one launch duration of the newest one
        t = t[t[:, 7] > t[:, 7].max() - int(us * 100 * 1.5)]
        ghz = lambda c0, c1: np.median((t[:, c1] - t[:, c0]) / np.maximum(t[:, c1 + 1] - t[:, c0 + 1], 1)) * 0.1
        seg = lambda c0, c1: float(np.median(t[:, c1 + 1] - t[:, c0 + 1])) * 1e-2
        clk_loop, clk_all = float(ghz(2, 4)), float(ghz(0, 6))
        peak = 256 * 4 * 1024 * clk_loop * 1e9 / 1e12
        rec = dict(case=name, us=round(us, 1), tflops=round(flops / us / 1e6, 0), workgroups=int(len(t)), loop_clock_ghz=round(clk_loop, 3),
                   kernel_clock_ghz=round(clk_all, 3), mfma_peak_at_loop_clock_tflops=round(peak, 0),
                   fill_us=round(seg(0, 2), 2), loop_us=round(seg(2, 4), 2), epilogue_us=round(seg(4, 6), 2), lifetime_us=round(seg(0, 6), 2))
        out.append(rec)
        print(json.dumps(rec), flush=True)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
