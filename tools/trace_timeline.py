#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: for the last full bench step, the time each kernel family
spends per queue, the union busy time, and the idle gaps on the main queue.

  python tools/trace_timeline.py gpurun_out/prof/x_kernel_trace.csv [--list]

--list: additionally every launch of that step in start order -- ms into the step, queue, duration, the idle time of the
union timeline in front of it (where nothing at all was running) and grid / workgroup size.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(gemm_tiled_kernel|gemm_dma_kernel|attn_fwd_kernel|attn_bwd_dq_kernel|attn_bwd_dkv_kernel|ln_fwd_kernel|"
                  r"ln_bwd_kernel|ln_fwd_vec_kernel|ln_bwd_vec_kernel|rows_cast_kernel|colsum_kernel|fold_bias_kernel|"
                  r"unfold_grads_kernel|embed_pos_fwd_kernel|embed_pos_bwd_kernel|pack_rows_\w+_kernel|gmu2_\w+_kernel|"
                  r"pack_weights_kernel|xblock_fwd_kernel|tail_\w+_kernel|im2col1d_kernel|col2im1d_kernel|pool_\w+_kernel|adam_kernel|"
                  r"adam_table_kernel|gemm_skinny_kernel|split_rows_kernel|add_n_kernel|zero_segments_kernel|signal_\w+_kernel)", name)
    if m:
        k = m.group(1)
        if k == "gemm_skinny_kernel":
            m2 = re.search(r"gemm_skinny_kernelI(?:DF16b|f)Lb([01])E", name)
            return k + ("<NT>" if m2 and m2.group(1) == "1" else "<NN>")
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
    # where nothing runs: gaps of the union timeline, largest first, with the kernels either side; and the low-occupancy
    # stretches (only kernels shorter than 30 us running)
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in step))
    gaps, cur_e, last = [], ev[0][1], ev[0][2]
    for st, en, k in ev[1:]:
        if st > cur_e:
            gaps.append((st - cur_e, 1e-6 * (cur_e - t0), last, k))
        if en > cur_e:
            cur_e, last = en, k
    hist = defaultdict(lambda: [0, 0])
    for g, _, a_, b_ in gaps:
        hist[(a_, b_)][0] += g
        hist[(a_, b_)][1] += 1
    print(f"{len(gaps)} gaps; by (kernel before -> kernel after), total us:")
    for (a_, b_), (g, n) in sorted(hist.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"   {a_:28s} -> {b_:28s} {n:4d} x {1e-3 * g / n:7.1f} us = {1e-3 * g:8.1f} us")
    print("largest single gaps (us, at ms into the step):")
    for g, at, a_, b_ in sorted(gaps, reverse=True)[:8]:
        print(f"   {1e-3 * g:8.1f} us at {at:7.3f} ms   {a_} -> {b_}")
    if "--list" in sys.argv:
        qn = {q: i for i, q in enumerate(sorted(per, key=lambda q: -qbusy[q]))}
        print("launches in start order: ms into the step | queue | us | idle in front (us) | kernel | grid / workgroup")
        cur_e = int(step[0]["Start_Timestamp"])
        for r in step:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            idle = max(0, st - cur_e)
            cur_e = max(cur_e, en)
            print(f"   {1e-6 * (st - t0):8.3f}  q{qn[r['Queue_Id']]}  {1e-3 * (en - st):8.1f}  {1e-3 * idle:6.1f}  {short(r['Kernel_Name']):40s} "
                  f"{r.get('Grid_Size_X', '?')}/{r.get('Workgroup_Size_X', '?')}")
    small = sum(en - st for st, en, k in ev if en - st < 30000)
    print(f"kernels shorter than 30 us: {sum(1 for st, en, k in ev if en - st < 30000)} launches, {1e-6 * small:.3f} ms of queue time")


if __name__ == "__main__":
    main()
