#!/usr/bin/env python3
"""End-to-end sanity on an MI355X: overfit ONE synthetic batch of the BASELINE shape with the HIP path + FusedAdam.
The loss must fall monotonically-ish and stay finite (dense and pruned schedules, bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["x"]
import bench  # noqa: E402
import bpmult_amd  # noqa: F401,E402
from bpmult_amd.models import get_model  # noqa: E402
from bpmult_amd.optim import FusedAdam  # noqa: E402

c = bench.CONFIGS["cfg1"]
dev = torch.device("cuda", 0)
for prune in (False, True):
    torch.manual_seed(0)
    model = get_model(bench.model_args(c, "bf16")).to(dev).train()
    model.set_prune_unused_rows(prune)
    batch = bench.synth_batch(c, c["batch"], 1234, dev)
    crit = torch.nn.BCEWithLogitsLoss()
    opt = FusedAdam(model, lr=2e-4)
    losses = []
    for it in range(60):
        opt.zero_grad()
        loss = crit(bench.run_model(model, batch), batch["tgt"])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    ok = all(map(lambda v: v == v and abs(v) < 1e3, losses)) and losses[-1] < 0.6 * losses[0]
    print(f"prune={prune}: loss {losses[0]:.4f} -> {losses[9]:.4f} -> {losses[29]:.4f} -> {losses[-1]:.4f}  {'OK' if ok else 'FAIL'}", flush=True)
    if not ok:
        sys.exit(1)
