#!/usr/bin/env python3
"""GEMM kernel lab (MI355X only): correctness of every LDS-DMA configuration against torch on edge shapes, then a
timing table of the encoder-layer GEMM shapes at hidden D for the register-staged kernel (cfg -2) and every LDS-DMA
configuration (cfg 0..), median of --iters launches on random data.

  python tools/gemm_lab.py [--d 768] [--rows 4096] [--G 6] [--check-only] [--time-only]

Uses the tuning hook bpm_debug_gemm_force of the -DBPM_LAB build (not in the product library or the C ABI header)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bpmult_amd  # noqa: E402,F401
from bpmult_amd import _lib, ops  # noqa: E402
from bpmult_amd.ops import (BPM_BF16, F_ACCUM, F_KPAD, F_RELU, GEMM_NN, GEMM_NT, GEMM_TN, OUT_CT, OUT_F32, OUT_HEADS)  # noqa: E402

DEV = "cuda"
CT = torch.bfloat16
NCFG = 7
NAMES = {-1: "auto", -2: "tiled", 0: "128x128 4w", 1: "256x128 8w", 2: "256x256 8w", 3: "256x256 16w", 4: "128x128 ns3", 5: "320x256 8w", 6: "256x128 2/CU"}


_LAB = None


def force(cfg):
    """The override lives in the -DBPM_LAB build only (build/lab/libbpmult_hip_lab.so, or BPMULT_LIB = a lab variant):
    this process's launches are routed through it from the first call on."""
    global _LAB
    if _LAB is None:
        if os.environ.get("BPMULT_LIB"):
            _LAB = _lib.lib()
            _LAB.bpm_debug_gemm_force.argtypes = [_lib.C.c_int]
        else:
            _LAB = _lib.lab_library().__enter__()
    _lib.check(_LAB.bpm_debug_gemm_force(cfg), "bpm_debug_gemm_force")


def pad(n, q=64):
    return (n + q - 1) // q * q


def rc(r, c, ld, scale=1.0):
    t = torch.zeros(r, ld, device=DEV, dtype=CT)
    t[:, :c] = (torch.randn(r, c, device=DEV) * scale).to(CT)
    return t


def check():
    bad = 0
    shapes = [(256, 256, 256), (300, 260, 320), (4096, 768, 768), (777, 1000, 1000), (512, 384, 192), (1600, 768, 3072),
              (8192, 2304, 640), (8000, 2052, 1500)]
    for variant, vn in ((GEMM_NT, "NT"), (GEMM_NN, "NN"), (GEMM_TN, "TN")):
        for (M, N, K) in shapes:
            if variant == GEMM_NT:
                A, B = rc(M, K, pad(K), 1.0), rc(N, K, pad(K), K ** -0.5)
                ref = A[:, :K].float() @ B[:, :K].float().T
            elif variant == GEMM_NN:
                A, B = rc(M, K, pad(K), 1.0), rc(K, N, pad(N, 8), K ** -0.5)
                ref = A[:, :K].float() @ B[:, :N].float()
            else:
                A, B = rc(K, M, pad(M, 8), 1.0), rc(K, N, pad(N, 8), K ** -0.5)
                ref = A[:, :M].float().T @ B[:, :N].float()
            bias = torch.randn(N, device=DEV)
            resid = torch.randn(M, N, device=DEV)
            ref2 = ref + bias + resid
            outs = {}
            for cfg in [-2] + list(range(NCFG)):
                force(cfg)
                out = torch.full((M, N), float("nan"), device=DEV)
                cs = torch.zeros(M, device=DEV)
                kw = dict(colsum_a=cs) if variant == GEMM_TN else {}
                p = ops.gemm_problem(A, B, out, M, N, K, A.shape[1], B.shape[1], N, bias_n=bias, resid=resid, ldr=N, flags=F_KPAD, **kw)
                ops.gemm_grouped(BPM_BF16, variant, [p], 5)
                torch.cuda.synchronize()
                err = (out - ref2).abs().max().item() / max(1.0, ref2.abs().max().item())
                ok = err < 2e-3 and torch.isfinite(out).all().item()
                if variant == GEMM_TN:
                    e2 = (cs - A[:, :M].float().sum(0)).abs().max().item() / max(1.0, A.float().sum(0).abs().max().item())
                    ok = ok and e2 < 2e-3
                outs[cfg] = out
                if cfg >= 0:
                    same = (out == outs[-2]).float().mean().item()
                else:
                    same = 1.0
                print(f"check {vn} M={M} N={N} K={K} cfg {cfg:2d} ({NAMES[cfg]:12s}): rel err {err:.2e} same-as-tiled {same:.3f} {'ok' if ok else 'FAIL'}", flush=True)
                bad += 0 if ok else 1
    # epilogues through the DMA kernel: relu + dropout -> CT, gate + colsum -> CT, heads scatter
    M, N, K = 1024, 768, 768
    B_, H, dh, dhp = 8, 6, 128, 128
    A, W = rc(M, K, K), rc(N, K, K, K ** -0.5)
    gate = rc(M, N, N)
    bias = torch.randn(N, device=DEV)
    res = {}
    for cfg in (-2, 0, 2, 3, 5):
        force(cfg)
        o1 = torch.full((M, N), float("nan"), device=DEV).to(CT)
        o2 = torch.full((M, N), float("nan"), device=DEV).to(CT)
        o3 = torch.zeros(B_, H, M // B_, dhp, device=DEV, dtype=CT)
        cs = torch.zeros(N, device=DEV)
        p1 = ops.gemm_problem(A, W, o1, M, N, K, K, K, N, bias_n=bias, flags=F_RELU | F_KPAD, drop_p=0.3, drop_site=5, out_kind=OUT_CT)
        p2 = ops.gemm_problem(A, W, o2, M, N, K, K, K, N, gate=gate, ldg=N, gate_scale=1.25, colsum=cs, flags=F_KPAD, out_kind=OUT_CT)
        p3 = ops.gemm_problem(A, W, o3, M, N, K, K, K, 0, bias_n=bias, alpha=0.2, out_kind=OUT_HEADS, heads=(B_, H, M // B_, dh, dhp), flags=F_KPAD)
        ops.gemm_grouped(BPM_BF16, GEMM_NT, [p1, p2, p3], seed=77)
        torch.cuda.synchronize()
        res[cfg] = (o1.float(), o2.float(), o3.float(), cs)
    for cfg in (0, 2, 3, 5):
        for i, nm in enumerate(("relu+drop CT", "gate CT", "heads", "colsum")):
            a, b = res[cfg][i], res[-2][i]
            err = (a - b).abs().max().item() / max(1.0, b.abs().max().item())
            ok = err < 1e-2
            print(f"check epilogue {nm:14s} cfg {cfg}: max rel diff vs tiled {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
            bad += 0 if ok else 1
    force(-1)
    return bad


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def bench(d, R, G, iters, B=8, H=12):
    ld, ld4 = d, 4 * d
    dh = d // H
    dhp = 32 if dh <= 32 else 64 if dh <= 64 else 128
    keep = []

    def case(name, variant, M, N, K, lda, ldb, ldc, nprob, **kw):
        probs = []
        for _ in range(nprob):
            if variant == GEMM_NT:
                A_, B_ = rc(M, K, lda, 0.5), rc(N, K, ldb, 0.5)
            elif variant == GEMM_NN:
                A_, B_ = rc(M, K, lda, 0.5), rc(K, N, ldb, 0.5)
            else:
                A_, B_ = rc(K, M, lda, 0.5), rc(K, N, ldb, 0.5)
            ok = kw.get("out_kind", OUT_F32)
            if ok == OUT_F32:
                C_ = torch.zeros(M, ldc, device=DEV)
            elif ok == OUT_CT:
                C_ = torch.zeros(M, ldc, device=DEV, dtype=CT)
            else:
                C_ = torch.zeros(B, H, M // B, dhp, device=DEV, dtype=CT)
            ex = {}
            if kw.get("bias"):
                ex["bias_n"] = torch.randn(N, device=DEV)
            if kw.get("resid"):
                ex["resid"], ex["ldr"] = torch.randn(M, N, device=DEV), N
            if kw.get("gate"):
                ex["gate"], ex["ldg"], ex["gate_scale"] = rc(M, N, ldc), ldc, 1.1
            if kw.get("colsum"):
                ex["colsum"] = torch.zeros(N, device=DEV)
            if kw.get("colsum_a"):
                ex["colsum_a"] = torch.zeros(M, device=DEV)
            if ok == OUT_HEADS:
                ex["heads"] = (B, H, M // B, dh, dhp)
            keep.extend([A_, B_, C_] + [v for v in ex.values() if torch.is_tensor(v)])
            probs.append(ops.gemm_problem(A_, B_, C_, M, N, K, lda, ldb, ldc, out_kind=ok, flags=kw.get("flags", 0) | F_KPAD,
                                          drop_p=0.0 if os.environ.get("LAB_NODROP") else kw.get("drop_p", 0.0), drop_site=3, **ex))
        arr = ops.array(ops.GemmProblem, probs)
        keep.append(arr)
        fl = 2.0 * M * N * K * nprob
        row = f"{name:32s}"
        for cfg in [-2] + list(range(NCFG)) + [-1]:
            force(cfg)
            ms = timeit(lambda: ops.gemm_grouped(BPM_BF16, variant, arr, 7), iters)
            row += f" | {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF"
        print(row, flush=True)

    print(f"d={d} rows={R} G={G}   columns: " + " | ".join(NAMES[c] for c in [-2] + list(range(NCFG)) + [-1]))
    case("NT 4096^3 plain f32 (1 problem)", GEMM_NT, 4096, 4096, 4096, 4096, 4096, 4096, 1)
    case("NT 8192x8192x4096 CT out", GEMM_NT, 8192, 8192, 4096, 4096, 4096, 8192, 1, out_kind=OUT_CT)
    if os.environ.get("LAB_SHORT"):
        case("NT out  (bias,resid,drop)", GEMM_NT, R, d, d, ld, ld, d, G, bias=True, resid=True, drop_p=0.1)
        case("NT fc1  (relu,drop,CT)", GEMM_NT, R, 4 * d, d, ld, ld, ld4, G, out_kind=OUT_CT, bias=True, flags=F_RELU, drop_p=0.1)
        case("NT fc2  (bias,resid,drop)", GEMM_NT, R, d, 4 * d, ld4, ld4, d, G, bias=True, resid=True, drop_p=0.1)
        case("TN wg ffn 2G (accum)", GEMM_TN, d, 4 * d, R, ld, ld4, 4 * d, 2 * G, flags=F_ACCUM)
        force(-1)
        return
    case("NT q    (heads,bias)", GEMM_NT, R, d, d, ld, ld, 0, G, out_kind=OUT_HEADS, bias=True)
    case("NT kv   (heads,bias) 2G", GEMM_NT, R, d, d, ld, ld, 0, 2 * G, out_kind=OUT_HEADS, bias=True)
    case("NT out  (bias,resid,drop)", GEMM_NT, R, d, d, ld, ld, d, G, bias=True, resid=True, drop_p=0.1)
    case("NT fc1  (relu,drop,CT)", GEMM_NT, R, 4 * d, d, ld, ld, ld4, G, out_kind=OUT_CT, bias=True, flags=F_RELU, drop_p=0.1)
    case("NT fc2  (bias,resid,drop)", GEMM_NT, R, d, 4 * d, ld4, ld4, d, G, bias=True, resid=True, drop_p=0.1)
    case("NN dfc2 (gate,colsum,CT)", GEMM_NN, R, 4 * d, d, ld, ld4, ld4, G, out_kind=OUT_CT, gate=True, colsum=True)
    case("NN dfc1 (f32)", GEMM_NN, R, d, 4 * d, ld4, ld, d, G)
    case("NN dout (heads)", GEMM_NN, R, d, d, ld, ld, 0, G, out_kind=OUT_HEADS)
    case("NN dq   (f32)", GEMM_NN, R, d, d, ld, ld, d, G)
    case("NN dkv  (f32 accum) 2G", GEMM_NN, R, d, d, ld, ld, d, 2 * G, flags=F_ACCUM)
    case("TN wg ffn 2G (accum)", GEMM_TN, d, 4 * d, R, ld, ld4, 4 * d, 2 * G, flags=F_ACCUM)
    case("TN wg att 4G (accum,colsum_a)", GEMM_TN, d, d, R, ld, ld, d, 3 * G, flags=F_ACCUM, colsum_a=True)
    force(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--G", type=int, default=6)
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--time-only", action="store_true")
    a = ap.parse_args()
    bad = 0
    if not a.time_only:
        bad = check()
        print(f"correctness: {'all ok' if bad == 0 else str(bad) + ' FAILED'}", flush=True)
    if not a.check_only and bad == 0:
        bench(a.d, a.rows, a.G, a.iters)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
