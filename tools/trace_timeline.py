#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: for the last full bench step, the time each kernel family
spends per queue, the union busy time, and the idle gaps on the main queue.

  python tools/trace_timeline.py gpurun_out/prof/x_kernel_trace.csv [--step-marker embed_pos_fwd]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(gemm_tiled_kernel|gemm_dma_kernel|attn_fwd_kernel|attn_bwd_dq_kernel|attn_bwd_dkv_kernel|ln_fwd_kernel|"
                  r"ln_bwd_kernel|ln_fwd_vec_kernel|ln_bwd_vec_kernel|rows_cast_kernel|colsum_kernel|fold_bias_kernel|"
                  r"unfold_grads_kernel|embed_pos_fwd_kernel|embed_pos_bwd_kernel|pack_rows_\w+_kernel|gmu2_\w+_kernel|"
                  r"pack_weights_kernel|xblock_fwd_kernel|tail_\w+_kernel|im2col1d_kernel|col2im1d_kernel|pool_\w+_kernel|adam_kernel)", name)
    if m:
        k = m.group(1)
        if k in ("gemm_tiled_kernel", "gemm_dma_kernel"):      # needs mangled names (rocprofv3 -M): the demangler garbles __bf16 templates
            m2 = re.search(r"gemm_tiled_kernelI(?:DF16b|f)Lb([01])ELb([01])E", name) or \
                re.search(r"gemm_tiled_kernel<[^,]+, (true|false), (true|false)", name) or \
                re.search(r"gemm_dma_kernelILb([01])ELb([01])E", name) or re.search(r"gemm_dma_kernel<(true|false), (true|false)", name)
            if m2:
                xk, yk = (g in ("1", "true") for g in m2.groups())
                k += "<NT>" if (xk and yk) else "<NN>" if xk else "<TN>"
            else:
                k += "<?>"
        return k
    if "Cijk" in name:
        return "hipblaslt"
    if "rccl" in name.lower() or "nccl" in name.lower():
        return "rccl"
    return "torch:" + name.split("(")[0][-40:]


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "embed_pos_fwd" in r["Kernel_Name"]]
    # two embed_pos_fwd launches per step (level 1, level 2): a step = marks[2k] .. marks[2k+2]
    if len(marks) < 6:
        sys.exit("not enough steps in trace")
    a, b = marks[-4], marks[-2]
    step = rows[a:b]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    print(f"step window {1e-6 * (t1 - t0):.3f} ms, {len(step)} kernels")
    per = defaultdict(lambda: defaultdict(lambda: [0, 0]))
    qbusy = defaultdict(int)
    for r in step:
        q = r["Queue_Id"]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        per[q][k][0] += d
        per[q][k][1] += 1
        qbusy[q] += d
    for q in sorted(per, key=lambda q: -qbusy[q]):
        print(f"queue {q}: busy {1e-6 * qbusy[q]:.3f} ms")
        for k, (d, n) in sorted(per[q].items(), key=lambda kv: -kv[1][0]):
            if d > 20000:
                print(f"   {k:44s} {n:5d} x {1e-3 * d / n:8.1f} us = {1e-6 * d:7.3f} ms")
    # union busy
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"union busy {1e-6 * busy:.3f} ms, idle {1e-6 * (t1 - t0 - busy):.3f} ms")


if __name__ == "__main__":
    main()
