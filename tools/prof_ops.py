import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.argv=["x"]
import bench, bpmult_amd
from bpmult_amd.models import get_model
c = bench.CONFIGS[os.environ.get("PROF_CFG", "h768")]; dev = torch.device("cuda", 0)
model = get_model(bench.model_args(c, "bf16")).to(dev).train()
batch = bench.synth_batch(c, c["batch"], 1234, dev)
crit = torch.nn.BCEWithLogitsLoss()
def step():
    for p in model.parameters(): p.grad = None
    loss = crit(bench.run_model(model, batch), batch["tgt"]); loss.backward(); return loss
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    step()
torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
for e in rows[:25]:
    print(f"{e.key[:50]:50s} count={e.count:5d} cpu_total={e.cpu_time_total/1e3:8.2f} ms")
