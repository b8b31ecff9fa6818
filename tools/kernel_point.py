#!/usr/bin/env python3
"""The north-star kernel point (BASELINE.json: "fused biprojection crossmodal-attention kernel at hidden_sz=768 /
seq_len=50"): one crossmodal attention block -- Q / K / V projections, masked softmax attention, output projection with
residual -- for six lock-stepped encoders at d=768, H=6 (head_dim 128), T=S=50, B=64, bf16, README dropout rates.

  python tools/kernel_point.py            (MI355X only; also imported by bench.py for its `kernel_point` object)

Timed stand-alone with HIP events (median of --iters launches, random data): the fused launch (bpm_xblock_fwd) and the
five separate launches it replaces (grouped Q/K/V projection GEMM, attention, output projection GEMM).  Algorithmic
FLOPs per sample and layer = (4T + 4S) d^2 + 4 P(T,S) d (P = visible (query, key) pairs)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK = 2500.0


def _time(fn, iters):
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
    return ts[len(ts) // 2] * 1e3


def measure(dev="cuda", d=768, H=6, T=50, S=50, B=64, G=6, iters=20):
    import bpmult_amd  # noqa: F401
    from bpmult_amd import ops
    from bpmult_amd.ops import BPM_BF16, F_KPAD, GEMM_NT, OUT_HEADS

    dh, ct = d // H, torch.bfloat16
    g = torch.Generator().manual_seed(5)
    rb = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(ct).to(dev)
    rf = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    off, scale = 1 + abs(S - T), dh ** -0.5
    keep, fused, qkv, att, outp = [], [], [], [], []
    for e in range(G):
        t = dict(xq=rb(T * B, d), xk=rb(S * B, d), xv=rb(S * B, d), Wq=rb(d, d, sc=d ** -0.5), Wk=rb(d, d, sc=d ** -0.5),
                 Wv=rb(d, d, sc=d ** -0.5), Wo=rb(d, d, sc=d ** -0.5), bq=rf(d, sc=0.1), bk=rf(d, sc=0.1), bv=rf(d, sc=0.1), bo=rf(d, sc=0.1),
                 resid=rf(T * B, d), out=torch.zeros(T * B, d, device=dev), qh=torch.zeros(B, H, T, dh, device=dev, dtype=ct),
                 kh=torch.zeros(B, H, S, dh, device=dev, dtype=ct), vh=torch.zeros(B, H, S, dh, device=dev, dtype=ct),
                 ao=torch.zeros(T * B, d, device=dev, dtype=ct), lse=torch.zeros(B, H, T, device=dev))
        keep.append(t)
        fused.append(ops.xblock_problem(t["xq"], t["xk"], t["xv"], t["Wq"], t["bq"], t["Wk"], t["bk"], t["Wv"], t["bv"], t["Wo"], t["bo"],
                                        t["resid"], t["out"], t["qh"], t["kh"], t["vh"], t["ao"], d, t["lse"], B, H, T, S, d, d, off, scale,
                                        attn_drop=0.1, attn_site=9 + 16 * e, res_drop=0.1, res_site=11 + 16 * e))
        for x, W, b_, dst, L, al in (("xq", "Wq", "bq", "qh", T, scale), ("xk", "Wk", "bk", "kh", S, 1.0), ("xv", "Wv", "bv", "vh", S, 1.0)):
            qkv.append(ops.gemm_problem(t[x], t[W], t[dst], L * B, d, d, d, d, 0, bias_n=t[b_], alpha=al, out_kind=OUT_HEADS,
                                        heads=(B, H, L, dh, dh), flags=F_KPAD))
        att.append(ops.attn_problem(t["qh"], t["kh"], t["vh"], t["ao"], d, t["lse"], B, H, T, S, dh, dh, off, drop_p=0.1, drop_site=9 + 16 * e))
        outp.append(ops.gemm_problem(t["ao"], t["Wo"], t["out"], T * B, d, d, d, d, d, bias_n=t["bo"], resid=t["resid"], ldr=d,
                                     drop_p=0.1, drop_site=11 + 16 * e, flags=F_KPAD))
    fa, qa, aa, oa = (ops.array(c, p) for c, p in ((ops._lib.XBlockProblem, fused), (ops.GemmProblem, qkv), (ops.AttnProblem, att),
                                                   (ops.GemmProblem, outp)))

    def separate():
        ops.gemm_grouped(BPM_BF16, GEMM_NT, qa, 3)
        ops.attn_fwd(BPM_BF16, aa, 3)
        ops.gemm_grouped(BPM_BF16, GEMM_NT, oa, 3)

    us_f = _time(lambda: ops.xblock_fwd(BPM_BF16, fa, 3), iters)
    us_s = _time(separate, iters)
    pairs = sum(min(S, t + off) for t in range(T))
    flops_sample = (4 * T + 4 * S) * d * d + 4 * pairs * d
    flops = flops_sample * B * G
    # algorithmic HBM bytes of the fused launch: bf16 sources in, fp32 residual in / out, bf16 weights once per encoder,
    # tensors saved for backward (q / k / v heads, attention output, LSE)
    hbm = G * (B * (T + 2 * S) * d * 2 + 2 * B * T * d * 4 + 4 * d * d * 2 + B * (T + 2 * S) * d * 2 + B * T * d * 2 + B * H * T * 4)
    tf = flops / us_f / 1e6
    return {"what": "forward crossmodal-attention block (Q/K/V projections + masked softmax attention + output projection + residual), "
                    "six encoders per launch, bf16, dropout .1/.1, stand-alone",
            "shape": {"d": d, "H": H, "head_dim": dh, "T": T, "S": S, "B": B, "encoders": G},
            "fused_us": round(us_f, 1), "separate_kernels_us": round(us_s, 1), "launches_replaced": 5,
            "flops_per_sample_layer": flops_sample, "flops_per_launch": flops, "tflops": round(tf, 1), "peak_tflops": PEAK,
            "frac": round(tf / PEAK, 4), "target_frac": 0.30,
            "algorithmic_hbm_bytes": hbm, "flop_per_byte": round(flops / hbm, 1),
            "tflops_separate": round(flops / us_s / 1e6, 1), "frac_separate": round(flops / us_s / 1e6 / PEAK, 4),
            "product_path": "separate kernels (engine.FUSE_SHORT_BLOCKS = False): the fused launch is correct but slower",
            "why_short_of_target": "98 % of the block's flops are four 768 x 768 projections per 50-row sequence; one 64-row workgroup per "
                                   "(encoder, batch element) needs 6 LDS-DMA instructions per 16 MFMAs and is paced by their issue cost "
                                   "(~150 cycles each), and 384 workgroups of 144 KB LDS run in two rounds on 256 CUs; the grouped GEMMs "
                                   "amortise the same weight bytes over 3200-row problems but stop at ~500 TFLOP/s for K = N = 768 "
                                   "(main loop and fp32 epilogue do not overlap: DESIGN.md section 5)"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--B", type=int, default=64)
    a = ap.parse_args()
    print(json.dumps(measure(iters=a.iters, B=a.B)))
