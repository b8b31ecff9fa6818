#!/usr/bin/env python3
"""Captured graphs must not be destroyed while the process goes on capturing / launching others (MI355X, torch 2.10 +
ROCm 7.0/7.2): reproduction and the evidence behind models/bpmult.py's policy (cap instead of LRU eviction, dropped trunks
park their graphs in _RETIRED_GRAPHS).  A host segfault kills the process, so every variant runs in a child process.

  python tools/graph_cache_probe.py            # every variant, one line each; writes gpurun_out/r04_graph_probe.json
  python tools/graph_cache_probe.py --child NAME

Variants
  park      the library's policy: three batch sizes in turn (MAX_TRUNKS = 2: every switch drops a trunk), the dropped
            trunk's graphs are parked, new trunks capture and replay                                   -> expected: ok
  destroy   the same with BPMULT_GRAPH_DESTROY=1: the dropped trunk's graphs are destroyed              -> crashes if the
            fault is in graph destruction
  torch     no code of this library: plain torch graphs (two streams joined by events, a few matmuls), destroy the
            oldest while newer ones live, capture more, replay all                                      -> tells whether the
            fault needs this library's kernels at all

The first version of this probe (LRU eviction inside one trunk, profiles/r04_graph_probe_eviction.json) crashed in 5 of
5 evicting variants (shared / per-key pool, with / without a device synchronise before the eviction, thread_local / global
capture mode) at the same step, and in 0 of 2 non-evicting ones.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

VARIANTS = {"park": {}, "destroy": {"BPMULT_GRAPH_DESTROY": "1"}, "torch": {}}


def child_trunks():
    import copy
    import torch
    import test_model_gpu as T
    m1 = T._toy()
    m2 = copy.deepcopy(m1)
    m1, m2 = m1.cuda().train(), m2.cuda().train()
    m1.use_graphs, m2.use_graphs = False, True
    lossf = torch.nn.functional.binary_cross_entropy_with_logits
    n = 0
    for rnd in range(3):
        for B in (2, 3, 1):
            x = T._toy_inputs(B=B, seed=20 + B)
            tgt = (torch.randn(B, 6, generator=torch.Generator().manual_seed(B)) > 0).float().cuda()
            for it in range(4):
                outs = []
                for m in (m1, m2):
                    for p in m.parameters():
                        p.grad = None
                    out = m(x[0], None, None, x[1], x[2])
                    lossf(out, tgt).backward()
                    outs.append(out.detach().clone())
                assert torch.equal(outs[0], outs[1]), (rnd, B, it)
                n += 1
                print(f"step {n} round {rnd} B {B} ok", flush=True)
    from bpmult_amd.models import bpmult as BM
    print("STATS retired", len(BM._RETIRED_GRAPHS), flush=True)


def child_torch():
    import torch
    dev = "cuda"
    side = torch.cuda.Stream()
    pool = torch.cuda.graph_pool_handle()

    def capture(n):
        a = torch.randn(n, n, device=dev)
        b = torch.randn(n, n, device=dev)
        (a @ b).relu(); (b @ a).tanh()          # the vendor GEMM picks its kernel / workspace outside the capture
        torch.cuda.synchronize()
        out = [None, None]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                out[1] = (a @ b).relu()
            out[0] = (b @ a).tanh()
            ev2 = torch.cuda.Event()
            ev2.record(side)
            main.wait_event(ev2)
            out[0] = out[0] + out[1]
        return dict(g=g, a=a, b=b, out=out[0])

    live, n = [], 0
    for rnd in range(12):
        live.append(capture(64 + 8 * rnd))
        live.append(capture(200 + 8 * rnd))
        if len(live) > 4:                       # destroy the two oldest while four newer ones live
            del live[0], live[0]
        for e in live:
            for _ in range(3):
                e["g"].replay()
            torch.cuda.synchronize()
            ref = (e["b"] @ e["a"]).tanh() + (e["a"] @ e["b"]).relu()
            assert torch.allclose(e["out"], ref, atol=1e-3), rnd
            n += 1
            print(f"step {n} round {rnd} ok", flush=True)
    print("STATS live", len(live), flush=True)


def main():
    if "--child" in sys.argv:
        (child_torch if sys.argv[sys.argv.index("--child") + 1] == "torch" else child_trunks)()
        return
    res = {}
    for name, env in VARIANTS.items():
        e = dict(os.environ)
        e.update(env)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name], env=e, capture_output=True, text=True, timeout=150)
        except subprocess.TimeoutExpired:
            res[name] = {"rc": "timeout"}
            print(name, "TIMEOUT -- stopping", flush=True)
            break
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith(("step", "STATS"))]
        res[name] = {"rc": r.returncode, "last": lines[-1] if lines else "", "env": env,
                     "fault": next((ln for ln in r.stderr.splitlines() if "Fatal" in ln or "Error" in ln), ""),
                     "stderr_tail": r.stderr.splitlines()[-3:] if r.returncode else []}
        print(name, res[name], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r04_graph_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
