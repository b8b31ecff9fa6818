#!/bin/bash
# A/B of environment switches on one box: tools/ab_env.sh OUT.txt "VAR=1 VAR2=x" "..." ...  ("-" = defaults).  Every variant runs
# `bench.py --steps 20 --warmup 6 --no-secondary --no-cpu-baseline` in its own process; the defaults run first and last.
OUT="$1"; shift
: > "$OUT"
for v in "-" "$@" "-"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  line=$(env $e timeout -k 10 240 python3 bench.py --steps 20 --warmup 6 --no-secondary --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | tail -1)
  ms=$(python3 -c "import json,sys; print(json.loads(sys.argv[1])['ms_per_step'])" "$line" 2>/dev/null || echo fail)
  echo "$v  ms_per_step=$ms" | tee -a "$OUT"
done
