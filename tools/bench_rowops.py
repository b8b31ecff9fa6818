#!/usr/bin/env python3
"""Micro-benchmark of the row kernels (LayerNorm fwd/bwd, rows_cast) at the model's shapes (MI355X only)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=300)
    ap.add_argument("--R", type=int, default=4096)
    ap.add_argument("--G", type=int, default=6)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import bpmult_amd  # noqa: F401
    from bpmult_amd import ops
    from bpmult_amd.ops import BPM_BF16, pad32

    d, R, G = a.d, a.R, a.G
    ld = pad32(d)
    dev = "cuda"
    rn = lambda *s: torch.randn(*s, device=dev)

    def timeit(name, fn, nbytes):
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
        print(f"{name:44s} {med * 1e3:8.1f} us  {nbytes / med / 1e6:8.1f} GB/s", flush=True)

    keep = []

    def ln_probs(n, gam=True, cast=False, add=True):
        ps = []
        for _ in range(n):
            x, g, mean, rstd, dy, addt, dx = rn(R, d), rn(d), rn(R), rn(R).abs(), rn(R, d), rn(R, d), rn(R, d)
            dg, db, cs = torch.zeros(d, device=dev), torch.zeros(d, device=dev), torch.zeros(d, device=dev)
            ct = torch.zeros(R, ld, device=dev, dtype=torch.bfloat16)
            keep.extend([x, g, mean, rstd, dy, addt, dx, dg, db, cs, ct])
            kw = dict(dgamma=dg, dbeta=db) if gam else {}
            if cast:
                kw.update(cast=ct, ldc=ld, cast_colsum=cs, drop_p=0.1, drop_site=3)
            ps.append(ops.ln_problem(x, g, None, mean, rstd, R, dy=dy, ldy=d, add=addt if add else None, dx=dx, **kw))
        return ops.array(ops.LnProblem, ps)

    base = R * d * 4 * 4
    for n in (G, 2 * G):
        for gam, cast in ((False, False), (True, False), (True, True)):
            arr = ln_probs(n, gam, cast)
            keep.append(arr)
            nb = n * (base + (R * ld * 2 if cast else 0))
            timeit(f"ln_bwd x{n} dgamma={int(gam)} cast={int(cast)}", lambda: ops.ln_bwd(arr, d, BPM_BF16, 5), nb)

    # ln fwd
    for n in (G, 2 * G):
        ps = []
        for _ in range(n):
            x, g, b, mean, rstd = rn(R, d), rn(d), rn(d), rn(R), rn(R)
            out = torch.zeros(R, ld, device=dev, dtype=torch.bfloat16)
            keep.extend([x, g, b, mean, rstd, out])
            ps.append(ops.ln_problem(x, g, b, mean, rstd, R, out=out, ldo=ld))
        arr2 = ops.array(ops.LnProblem, ps)
        keep.append(arr2)
        timeit(f"ln_fwd x{n}", lambda: ops.ln_fwd(BPM_BF16, arr2, d), n * (R * d * 4 + R * ld * 2))

    # colsum of CT rows (in_proj bias gradients)
    ps = []
    for _ in range(3 * G):
        src = torch.randn(R, ld, device=dev).to(torch.bfloat16)
        cs = torch.zeros(d, device=dev)
        keep.extend([src, cs])
        ps.append(ops.cast_problem(src, ld, R, d, a_is_ct=True, colsum=cs))
    arr3 = ops.array(ops.CastProblem, ps)
    timeit(f"rows_cast colsum-only x{3 * G}", lambda: ops.rows_cast(BPM_BF16, arr3, 5), 3 * G * R * ld * 2)
    # f32 -> CT cast with dropout + colsum
    ps = []
    for _ in range(G):
        src, cs = rn(R, d), torch.zeros(d, device=dev)
        dst = torch.zeros(R, ld, device=dev, dtype=torch.bfloat16)
        keep.extend([src, cs, dst])
        ps.append(ops.cast_problem(src, d, R, d, dst_ct=dst, ldd=ld, colsum=cs, drop_p=0.1, drop_site=4))
    arr4 = ops.array(ops.CastProblem, ps)
    timeit(f"rows_cast f32->CT drop colsum x{G}", lambda: ops.rows_cast(BPM_BF16, arr4, 5), G * (R * d * 4 + R * ld * 2))


if __name__ == "__main__":
    main()
