import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
import bench
from types import SimpleNamespace
import bpmult_amd
from bpmult_amd.models import get_model
c = bench.CONFIGS["cfg1"]
dev = torch.device("cuda", 0)
model = get_model(bench.model_args(c, "bf16")).to(dev).train()
batch = bench.synth_batch(c, c["batch"], 1234, dev)
crit = torch.nn.BCEWithLogitsLoss()
def step():
    for p in model.parameters():
        p.grad = None
    loss = crit(bench.run_model(model, batch), batch["tgt"])
    loss.backward()
    return loss
for _ in range(5):
    step()
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(t1 - t0); ts.append(t2 - t0)
print("host enqueue ms", sorted(hs)[5] * 1e3, "total ms", sorted(ts)[5] * 1e3)
