#!/bin/bash
# Instruction counters of the three attention kernels on the stand-alone launches of tools/attn_lab.py (run on the GPU box from
# the repo root): tools/attn_pmc.sh OUT.json     (two counter passes; --pmc runs carry --kernel-trace only)
set -e
OUT="$1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/attn_pmc_tmp
rm -rf $D && mkdir -p $D
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $D/p$i -o p -- python3 tools/attn_lab.py --iters 3 > $D/p$i.log 2>&1
done
python3 - "$D" "$OUT" <<'PY'
import csv, glob, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(sys.argv[1] + "/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" not in k:
            continue
        a = acc[k[:40]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {k: dict({"launches": max(n for _, n in v.values())}, **{c: round(s / n) for c, (s, n) in sorted(v.items())}) for k, v in acc.items()}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out)[:600])
PY
rm -rf $D
