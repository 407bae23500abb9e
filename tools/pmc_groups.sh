#!/bin/bash
# PMC passes over the gather-mm kernels (tools/bench_conv.py, level given as $1, default 1): each group of counters is
# its own rocprofv3 run with --kernel-trace only (the pool refuses --pmc together with the hip/hsa trace domains).
# usage: bash tools/pmc_groups.sh [level] [outdir]     -> <outdir>/g<i>/...counter_collection.csv, summarised by pmc_summary.py
LEVEL=${1:-1}
OUT=${2:-gpurun_out/pmc_tile4}
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  # shellcheck disable=SC2086
  timeout 300 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/g$i" -- python3 tools/bench_conv.py --levels "$LEVEL" --variants 4 > "$OUT/g$i.log" 2>&1
done <<'GROUPS'
GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES
TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCR_TCP_STALL_CYCLES GRBM_TA_BUSY
GROUPS
