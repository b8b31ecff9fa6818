#!/usr/bin/env python3
"""Average the counters collected by tools/pmc_gemm.sh per kernel name."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm_tiled"
acc = defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(f"{out}/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k:44s} {v / n:16.1f}   (n={n})")
