#!/usr/bin/env python3
"""Per-kernel-family HBM traffic per launch from the two PMC passes of tools/pmc_traffic.sh."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from trace_timeline import short  # noqa: E402

root, out = sys.argv[1], sys.argv[2]
acc = {"FETCH_SIZE": defaultdict(lambda: [0.0, 0]), "WRITE_SIZE": defaultdict(lambda: [0.0, 0])}
for f in glob.glob(f"{root}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        c = r["Counter_Name"]
        if c in acc:
            a = acc[c][short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
res = {}
for k in sorted(set(acc["FETCH_SIZE"]) | set(acc["WRITE_SIZE"])):
    f, nf = acc["FETCH_SIZE"].get(k, [0.0, 0])
    w, nw = acc["WRITE_SIZE"].get(k, [0.0, 0])
    if not nf or not nw or k.startswith("torch:"):
        continue
    # counters are reported in KB; gfx950: double FETCH_SIZE (128-B requests tallied at 64 B)
    res[k] = {"launches": nf, "fetch_bytes_per_launch": 2.0 * 1024.0 * f / nf, "write_bytes_per_launch": 1024.0 * w / nw,
              "hbm_bytes_per_launch": 2.0 * 1024.0 * f / nf + 1024.0 * w / nw}
def csrc_sha16():
    """Hash of the kernel sources the measured library was built from: bench.py marks the figure stale when they differ."""
    base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "biprojection-multimodal-transformer_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(base)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(base, f), "rb").read())
    return h.hexdigest()[:16]


json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 3 --warmup 1", "csrc_sha16": csrc_sha16(),
           "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)", "kernels": res}, open(out, "w"), indent=1)
for k, v in res.items():
    print(f"{k:40s} n={v['launches']:5d}  fetch {v['fetch_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")
