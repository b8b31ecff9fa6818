#!/usr/bin/env python3
"""Host enqueue time of one bench step against its GPU time (MI355X only): if the host needs about as long as the GPU,
the step is host-bound at its boundaries (the GPU idles until the next forward's first kernel arrives).

  python tools/host_time.py [--config h768|cfg1|cfg3|...] [--graphs 0|1]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="h768")
ap.add_argument("--graphs", type=int, default=1)
a = ap.parse_args()
c = bench.CONFIGS[a.config]
dev = torch.device("cuda", 0)
model = bench.make_model(c, "bf16").to(dev).train()
model.use_graphs = bool(a.graphs)
batch = bench.synth_batch(c, c["batch"], 1234, dev)
crit = torch.nn.BCEWithLogitsLoss()
T = {"zero_grad": [], "forward": [], "loss": [], "backward": [], "host_total": [], "gpu_total": []}


def step(rec=False):
    t0 = time.perf_counter()
    for p in model.parameters():
        p.grad = None
    t1 = time.perf_counter()
    out = bench.run_model(model, batch)
    t2 = time.perf_counter()
    loss = crit(out, batch["tgt"])
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    if rec:
        for k, v in (("zero_grad", t1 - t0), ("forward", t2 - t1), ("loss", t3 - t2), ("backward", t4 - t3), ("host_total", t4 - t0)):
            T[k].append(v * 1e3)
    return loss


for _ in range(6):
    step()
torch.cuda.synchronize()
# (a) host alone: synchronise before each step so the host never waits for queue space
for _ in range(8):
    torch.cuda.synchronize()
    step(True)
torch.cuda.synchronize()
# (b) steady state: back-to-back steps
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
steady = (time.perf_counter() - t0) / n * 1e3
med = lambda v: sorted(v)[len(v) // 2]
print(json.dumps({"config": a.config, "graphs": a.graphs, "host_ms": {k: round(med(v), 3) for k, v in T.items() if v},
                  "steady_state_ms_per_step": round(steady, 3),
                  "n_parameters": sum(1 for _ in model.parameters())}))
