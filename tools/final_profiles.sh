#!/bin/bash
# Final-build evidence for profiles/ (run on the GPU box from the repo root): kernel stats + timeline, HBM traffic (PMC),
# kernel point under the profiler, stand-alone GEMM / attention tables.   tools/final_profiles.sh r03_k
set -e
TAG="$1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rm -rf $O/prof_$TAG && mkdir -p $O/prof_$TAG
rocprofv3 -M --kernel-trace --stats --output-format csv -d $O/prof_$TAG/k -o k -- python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-secondary > $O/${TAG}_prof.log 2>&1
python3 tools/trace_timeline.py $O/prof_$TAG/k/k_kernel_trace.csv > $O/${TAG}_timeline_h768.txt
cp $O/prof_$TAG/k/k_kernel_stats.csv $O/${TAG}_kernel_stats_h768.csv
rm -f $O/prof_$TAG/k/k_kernel_trace.csv
echo "== timeline"; cat $O/${TAG}_timeline_h768.txt
bash tools/pmc_traffic.sh $O/${TAG}_hbm_traffic_h768.json > $O/${TAG}_traffic.log 2>&1
echo "== traffic"; tail -12 $O/${TAG}_traffic.log
rocprofv3 -M --kernel-trace --stats --output-format csv -d $O/prof_$TAG/kp -o kp -- python3 tools/kernel_point.py > $O/${TAG}_kernel_point.json 2> $O/${TAG}_kp.err
cp $O/prof_$TAG/kp/kp_kernel_stats.csv $O/${TAG}_kernel_point_stats.csv
rm -f $O/prof_$TAG/kp/kp_kernel_trace.csv
echo "== kernel point"; cut -c1-700 $O/${TAG}_kernel_point.json
python3 tools/attn_lab.py > $O/${TAG}_attn_lab_h768.json 2>/dev/null || true
python3 tools/gemm_lab.py --time-only > $O/${TAG}_gemm_lab_h768.txt 2>&1 || true
bash tools/attn_pmc.sh $O/${TAG}_attn_pmc.json > $O/${TAG}_attn_pmc.log 2>&1 || true
rm -rf $O/prof_$TAG
echo "== done"
