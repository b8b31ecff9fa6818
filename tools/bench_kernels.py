#!/usr/bin/env python3
"""Micro-benchmark of the grouped kernels at the model's shapes (MI355X only).

  python tools/bench_kernels.py [--d 300] [--B 8] [--N 512] [--G 6] [--lib path/to/alt.so]

Times each launch shape of one encoder-group layer step with HIP events on the
current stream (median of --iters back-to-back launches, random data) and
prints algorithmic TFLOP/s.  Used to iterate on kernel variants; not a test.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=300)
    ap.add_argument("--H", type=int, default=12)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--N", type=int, default=512)
    ap.add_argument("--G", type=int, default=6)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    a = ap.parse_args()

    import bpmult_amd  # noqa: F401
    from bpmult_amd import ops
    from bpmult_amd.ops import (BPM_BF16, BPM_F32, F_ATOMIC, F_RELU, GEMM_NN, GEMM_NT, GEMM_TN, OUT_CT, OUT_F32, OUT_HEADS, pad32)

    dt = BPM_BF16 if a.dtype == "bf16" else BPM_F32
    ct = ops.ct_torch(dt)
    d, B, N, G, H = a.d, a.B, a.N, a.G, a.H
    R = N * B
    ld, ld4 = pad32(d), pad32(4 * d)
    dh = d // H
    dhp = 32 if dh <= 32 else 64 if dh <= 64 else 128
    dev = "cuda"
    rn = lambda *s: torch.randn(*s, device=dev)
    rc = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(ct)

    def timeit(name, fn, flops):
        if a.only and a.only not in name:
            return
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{name:34s} {med * 1e3:9.1f} us  {flops / med / 1e9:8.1f} TFLOP/s   (min {ts[0] * 1e3:.1f} us)", flush=True)

    keep = []

    def gemm_case(name, variant, M, Nn, K, lda, ldb, ldc, nprob, **kw):
        probs = []
        for _ in range(nprob):
            if variant == GEMM_NT:
                A_, B_ = rc(M, lda), rc(Nn, ldb)
            elif variant == GEMM_NN:
                A_, B_ = rc(M, lda), rc(K, ldb)
            else:
                A_, B_ = rc(K, lda), rc(K, ldb)
            out_kind = kw.get("out_kind", OUT_F32)
            if out_kind == OUT_F32:
                C_ = torch.zeros(M, ldc, device=dev)
            elif out_kind == OUT_CT:
                C_ = torch.zeros(M, ldc, device=dev, dtype=ct)
            else:
                C_ = torch.zeros(B, H, M // B, dhp, device=dev, dtype=ct)
            extra = {}
            if kw.get("bias"):
                extra["bias_n"] = rn(Nn)
            if kw.get("resid"):
                extra["resid"], extra["ldr"] = rn(M, Nn), Nn
            if kw.get("gate"):
                extra["gate"], extra["ldg"], extra["gate_scale"] = rc(M, ldc), ldc, 1.1
            if kw.get("colsum"):
                extra["colsum"] = torch.zeros(Nn, device=dev)
            if out_kind == OUT_HEADS:
                extra["heads"] = (B, H, M // B, dh, dhp)
            keep.extend([A_, B_, C_] + list(extra.values()))
            probs.append(ops.gemm_problem(A_, B_, C_, M, Nn, K, lda, ldb, ldc, out_kind=out_kind, flags=kw.get("flags", 0) | (0 if os.environ.get("NO_KPAD") else 8),
                                          drop_p=kw.get("drop_p", 0.0), drop_site=3, splitk=kw.get("splitk", 1), **extra))
        arr = ops.array(ops.GemmProblem, probs)
        keep.append(arr)
        timeit(name, lambda: ops.gemm_grouped(dt, variant, arr, 7), 2.0 * M * Nn * K * nprob)

    # ---- forward
    gemm_case("NT qkv   (heads, bias)", GEMM_NT, R, d, d, ld, ld, 0, 3 * G, out_kind=OUT_HEADS, bias=True)
    gemm_case("NT out   (bias,resid,drop)", GEMM_NT, R, d, d, ld, ld, d, G, bias=True, resid=True, drop_p=0.1)
    gemm_case("NT out   (plain f32)", GEMM_NT, R, d, d, ld, ld, d, G)
    gemm_case("NT out   (bias,resid) nodrop", GEMM_NT, R, d, d, ld, ld, d, G, bias=True, resid=True)
    gemm_case("NT fc1   (relu,CT) nodrop", GEMM_NT, R, 4 * d, d, ld, ld, ld4, G, out_kind=OUT_CT, bias=True, flags=F_RELU)
    gemm_case("NT fc1   (relu,drop,CT)", GEMM_NT, R, 4 * d, d, ld, ld, ld4, G, out_kind=OUT_CT, bias=True, flags=F_RELU, drop_p=0.1)
    gemm_case("NT fc2   (bias,resid,drop)", GEMM_NT, R, d, 4 * d, ld4, ld4, d, G, bias=True, resid=True, drop_p=0.1)
    # ---- backward
    gemm_case("NN dgrad fc2 (gate,colsum,CT)", GEMM_NN, R, 4 * d, d, ld, ld4, ld4, G, out_kind=OUT_CT, gate=True, colsum=True)
    gemm_case("NN dgrad fc1 (f32)", GEMM_NN, R, d, 4 * d, ld4, ld, d, G)
    gemm_case("NN dgrad out (heads)", GEMM_NN, R, d, d, ld, ld, 0, G, out_kind=OUT_HEADS)
    gemm_case("NN dgrad qkv (f32)", GEMM_NN, R, d, d, ld, ld, d, 3 * G)
    from bpmult_amd.ops import F_ACCUM
    gemm_case("TN wgrad ffn 2G (accum)", GEMM_TN, d, 4 * d, R, ld, ld4, 4 * d, 2 * G, flags=F_ACCUM)
    gemm_case("TN wgrad ffn 2G (atomic splitk2)", GEMM_TN, d, 4 * d, R, ld, ld4, 4 * d, 2 * G, flags=F_ATOMIC, splitk=2)
    gemm_case("TN wgrad att 4G (accum)", GEMM_TN, d, d, R, ld, ld, d, 4 * G, flags=F_ACCUM)
    gemm_case("TN wgrad att 4G (atomic splitk2)", GEMM_TN, d, d, R, ld, ld, d, 4 * G, flags=F_ATOMIC, splitk=2)
    gemm_case("TN wgrad att 4G (atomic splitk4)", GEMM_TN, d, d, R, ld, ld, d, 4 * G, flags=F_ATOMIC, splitk=4)

    # ---- attention (causal T = S = N)
    aps = []
    for _ in range(G):
        Q, K_, V = rc(B, H, N, dhp), rc(B, H, N, dhp), rc(B, H, N, dhp)
        O = torch.zeros(R, ld, device=dev, dtype=ct)
        lse, delta = torch.zeros(B, H, N, device=dev), torch.zeros(B, H, N, device=dev)
        dO = rc(B, H, N, dhp)
        dQ, dK, dV = (torch.zeros(R, ld, device=dev, dtype=ct) for _ in range(3))
        keep.extend([Q, K_, V, O, lse, delta, dO, dQ, dK, dV])
        aps.append(ops.attn_problem(Q, K_, V, O, ld, lse, B, H, N, N, dh, dhp, 1, dO=dO, delta=delta, dQ=dQ, lddq=ld, dK=dK, lddk=ld,
                                    dV=dV, lddv=ld, dq_scale=dh ** -0.5, drop_p=float(os.environ.get("ATTN_DROP", "0.1")), drop_site=5))
    arr = ops.array(ops.AttnProblem, aps)
    pairs = N * (N + 1) / 2
    timeit("attn fwd (causal, drop)", lambda: ops.attn_fwd(dt, arr, 3), 4.0 * pairs * dh * B * H * G)
    timeit("attn bwd (dq + dkv)", lambda: ops.attn_bwd(dt, arr, 3), 8.0 * pairs * dh * B * H * G)


if __name__ == "__main__":
    main()
