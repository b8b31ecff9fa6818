#!/usr/bin/env python3
"""L2 behaviour of one GEMM launch under rocprofv3 PMC (run from the repo root on the GPU box):

  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d gpurun_out/l2 -o l2 -- python3 tools/gemm_l2_probe.py tn

Cases: tn = FFN weight gradients (12 problems 768 x 3072 x 4096), fc1 = NT 4096 x 3072 x 768 x 6, big = NT 4096^3.
Five launches of the case, nothing else on the GPU."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bpmult_amd  # noqa: F401,E402
from bpmult_amd import ops  # noqa: E402
from bpmult_amd.ops import BPM_BF16, F_ACCUM, F_KPAD, F_RELU, GEMM_NT, GEMM_TN, OUT_CT, OUT_F32  # noqa: E402

DEV, CT = "cuda", torch.bfloat16
rc = lambda r, c, s=0.5: (torch.randn(r, c, device=DEV) * s).to(CT)
case = sys.argv[1] if len(sys.argv) > 1 else "tn"
d, R, G = 768, 4096, 6
probs, keep = [], []
if case == "tn":
    for i in range(2 * G):
        M, N = (d, 4 * d) if i % 2 == 0 else (4 * d, d)
        A_, B_, C_ = rc(R, M), rc(R, N), torch.zeros(M, N, device=DEV)
        keep += [A_, B_, C_]
        probs.append(ops.gemm_problem(A_, B_, C_, M, N, R, M, N, N, flags=F_ACCUM | F_KPAD))
    variant = GEMM_TN
elif case == "fc1":
    for _ in range(G):
        A_, B_, C_ = rc(R, d), rc(4 * d, d, d ** -0.5), torch.zeros(R, 4 * d, device=DEV, dtype=CT)
        bias = torch.randn(4 * d, device=DEV)
        keep += [A_, B_, C_, bias]
        probs.append(ops.gemm_problem(A_, B_, C_, R, 4 * d, d, d, d, 4 * d, out_kind=OUT_CT, bias_n=bias, flags=F_RELU | F_KPAD, drop_p=0.1, drop_site=3))
    variant = GEMM_NT
else:
    A_, B_, C_ = rc(4096, 4096), rc(4096, 4096, 1 / 64), torch.zeros(4096, 4096, device=DEV)
    keep += [A_, B_, C_]
    probs.append(ops.gemm_problem(A_, B_, C_, 4096, 4096, 4096, 4096, 4096, 4096, out_kind=OUT_F32, flags=F_KPAD))
    variant = GEMM_NT
arr = ops.array(ops.GemmProblem, probs)
for _ in range(5):
    ops.gemm_grouped(BPM_BF16, variant, arr, 7)
torch.cuda.synchronize()
print("done", case)
