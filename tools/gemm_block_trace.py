#!/usr/bin/env python3
"""Per-workgroup phase timeline of the tiled GEMM (needs a -DBPM_GEMM_TRACE build: BPMULT_LIB=variants/lib_trace.so)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bpmult_amd  # noqa: F401,E402
from bpmult_amd import _lib, ops  # noqa: E402
from bpmult_amd.ops import BPM_BF16, F_RELU, GEMM_NT, OUT_CT, pad32  # noqa: E402

d, R, G = 300, 4096, 6
ld, ld4 = pad32(d), pad32(4 * d)
dev = "cuda"
rc = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
keep, probs = [], []
which = sys.argv[1] if len(sys.argv) > 1 else "fc1"
for _ in range(G):
    if which == "fc1":
        A_, B_, C_ = rc(R, ld), rc(4 * d, ld), torch.zeros(R, ld4, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(4 * d, device=dev)
        keep += [A_, B_, C_, bias]
        probs.append(ops.gemm_problem(A_, B_, C_, R, 4 * d, d, ld, ld, ld4, out_kind=OUT_CT, flags=F_RELU, drop_p=0.1, drop_site=3, bias_n=bias))
    elif which == "wgrad":   # dW[4d, d] = dh1^T xn2 : TN, K = R
        A_, B_, C_ = rc(R, ld4), rc(R, ld), torch.zeros(4 * d, d, device=dev)
        keep += [A_, B_, C_]
        probs.append(ops.gemm_problem(A_, B_, C_, 4 * d, d, R, ld4, ld, d, flags=ops.F_ACCUM))
        A2, B2, C2 = rc(R, ld), rc(R, ld4), torch.zeros(d, 4 * d, device=dev)
        keep += [A2, B2, C2]
        probs.append(ops.gemm_problem(A2, B2, C2, d, 4 * d, R, ld, ld4, 4 * d, flags=ops.F_ACCUM))
    else:   # fc2
        A_, B_, C_ = rc(R, ld4), rc(d, ld4), torch.zeros(R, d, device=dev)
        bias, res = torch.randn(d, device=dev), torch.randn(R, d, device=dev)
        keep += [A_, B_, C_, bias, res]
        probs.append(ops.gemm_problem(A_, B_, C_, R, d, 4 * d, ld4, ld4, d, drop_p=0.1, drop_site=3, bias_n=bias, resid=res, ldr=d))
arr = ops.array(ops.GemmProblem, probs)
for _ in range(3):
    ops.gemm_grouped(BPM_BF16, ops.GEMM_TN if which == "wgrad" else GEMM_NT, arr, 7)
torch.cuda.synchronize()
lib = _lib.lib()
lib.bpm_debug_trace.argtypes = [C.c_void_p, C.c_int]
nb = 8192
buf = np.zeros((nb, 16), dtype=np.uint64)
assert lib.bpm_debug_trace(buf.ctypes.data, nb) == 0
used = buf[:, 0] > 0
t = buf[used].astype(np.int64)
t0 = t[:, 0].min()
print("blocks traced", used.sum())
start = (t[:, 0] - t0) * 10e-3          # us
end = (t[:, 15] - t0) * 10e-3
print(f"kernel span {end.max():.1f} us; block lifetime mean {np.mean(end - start):.2f} us  p10 {np.percentile(end - start, 10):.2f}  p90 {np.percentile(end - start, 90):.2f}")
pro = (t[:, 1] - t[:, 0]) * 10e-3
nk = int(((t[:, 2:12] > 0).sum(1)).max())
loop = (t[:, 1 + nk] - t[:, 1]) * 10e-3
epi = (t[:, 15] - t[:, 1 + nk]) * 10e-3
print(f"prologue {pro.mean():.2f} us | k-loop ({nk} it) {loop.mean():.2f} us = {loop.mean() / nk:.2f}/it | epilogue {epi.mean():.2f} us")
if (t[:, 12] > 0).all():
    seg = lambda a, b: ((t[:, b] - t[:, a]) * 10e-3).mean()
    print(f"epilogue: rows {seg(1 + nk, 12):.2f} | cols a=0 {seg(12, 13):.2f} | cols a=1 {seg(13, 14):.2f} | drain {seg(14, 15):.2f}")
# concurrency profile
ev = sorted([(s, 1) for s in start] + [(e, -1) for e in end])
cur, peak = 0, 0
for _, dlt in ev:
    cur += dlt
    peak = max(peak, cur)
print("peak concurrent blocks", peak)
for lo in range(0, int(end.max()) + 1, 10):
    n = ((start < lo + 10) & (end > lo)).sum()
    print(f"  t={lo:4d}us active~{n}")
