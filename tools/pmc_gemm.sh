#!/bin/bash
# Hardware-counter passes over one microbenchmark shape (run on the GPU box from the repo root).
#   tools/pmc_gemm.sh "NT fc1" outdir
set -e
ONLY="$1"; OUT="$2"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES TA_BUSY_avr SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o p -- python3 tools/bench_kernels.py --only "$ONLY" --iters 3 > "$OUT/p$i.log" 2>&1
done
