#!/bin/bash
# HBM traffic per launch of every kernel family of the bench step (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots:
# FETCH_SIZE and WRITE_SIZE in separate passes; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2).
#   [BENCH_ARGS="--config cfg1"] tools/pmc_traffic.sh out.json        (run on the GPU box from the repo root; default workload h768)
set -e
OUT="$1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_tr && mkdir -p /tmp/pmc_tr
rocprofv3 -M --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_tr/f -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ${BENCH_ARGS} > /tmp/pmc_tr/f.log 2>&1
rocprofv3 -M --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_tr/w -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ${BENCH_ARGS} > /tmp/pmc_tr/w.log 2>&1
python3 tools/pmc_traffic_summary.py /tmp/pmc_tr "$OUT"
