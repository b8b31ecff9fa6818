#!/usr/bin/env python3
"""Attention kernel lab (MI355X only): the three attention launches of one encoder layer at the headline workload's
shapes -- six lock-stepped encoders, B=8, H=12 heads of 64, T=S=512, future mask, attention dropout .1 on two of the six
(README rates) -- timed stand-alone with HIP events (median of --iters launches, random data).

  python tools/attn_lab.py [--H 12 --dh 64 --T 512 --S 512 --B 8 --G 6 --no-mask --drop 0.1]
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES ... -- python3 tools/attn_lab.py --iters 3     (counters per kernel)

Algorithmic FLOPs per visible (query, key) pair and head: forward 4 dh, dQ pass 6 dh (S, dP, dQ), dK/dV pass 8 dh."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bpmult_amd  # noqa: E402,F401
from bpmult_amd import ops  # noqa: E402
from bpmult_amd.ops import BPM_BF16  # noqa: E402


def _time(fn, iters):
    for _ in range(2):
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


def main():
    ap = argparse.ArgumentParser()
    for k, v in (("H", 12), ("dh", 64), ("T", 512), ("S", 512), ("B", 8), ("G", 6), ("iters", 20)):
        ap.add_argument("--" + k, type=int, default=v)
    ap.add_argument("--drop", type=float, default=0.1)
    ap.add_argument("--drop-encoders", type=int, default=2, help="how many of the G encoders have attention dropout")
    ap.add_argument("--no-mask", action="store_true")
    ap.add_argument("--pair", type=int, default=-1, help="tuning hook: bit k = kernel k (fwd, dQ, dK/dV) takes two blocks per workgroup")
    a = ap.parse_args()
    dev, ct = "cuda", torch.bfloat16
    if a.pair >= 0:
        from bpmult_amd import _lib
        _lib.check(_lib.lab_library().__enter__().bpm_debug_attn_pair(a.pair), "bpm_debug_attn_pair")   # -DBPM_LAB build only
    B, H, T, S, dh, G = a.B, a.H, a.T, a.S, a.dh, a.G
    dhp = 32 if dh <= 32 else 64 if dh <= 64 else 128
    d = H * dh
    off = (1 << 30) if a.no_mask else 1 + abs(S - T)
    g = torch.Generator().manual_seed(3)
    rb = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(ct).to(dev)
    keep, probs = [], []
    for e in range(G):
        t = dict(q=rb(B, H, T, dhp), k=rb(B, H, S, dhp), v=rb(B, H, S, dhp), do=rb(B, H, T, dhp),
                 o=torch.zeros(T * B, d, device=dev, dtype=ct), lse=torch.zeros(B, H, T, device=dev),
                 delta=torch.zeros(B, H, T, device=dev), dq=torch.zeros(T * B, d, device=dev, dtype=ct),
                 dk=torch.zeros(S * B, d, device=dev, dtype=ct), dv=torch.zeros(S * B, d, device=dev, dtype=ct))
        keep.append(t)
        probs.append(ops.attn_problem(t["q"], t["k"], t["v"], t["o"], d, t["lse"], B, H, T, S, dh, dhp, off, dO=t["do"], delta=t["delta"],
                                      dQ=t["dq"], lddq=d, dK=t["dk"], lddk=d, dV=t["dv"], lddv=d, dq_scale=dh ** -0.5,
                                      drop_p=a.drop if e < a.drop_encoders else 0.0, drop_site=9 + 16 * e))
    arr = ops.array(ops.AttnProblem, probs)
    pairs = sum(min(S, t + off) for t in range(T)) * B * H * G
    res = {"shape": dict(B=B, H=H, T=T, S=S, head_dim=dh, encoders=G, mask=not a.no_mask, drop=a.drop, drop_encoders=a.drop_encoders)}
    for name, fn, fl in (("fwd", ops.attn_fwd, 4), ("bwd_dq", ops.attn_bwd_dq, 6), ("bwd_dkv", ops.attn_bwd_dkv, 8)):
        us = _time(lambda: fn(BPM_BF16, arr, 5), a.iters)
        res[name] = {"us": round(us, 1), "tflops": round(fl * dh * pairs / us / 1e6, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
