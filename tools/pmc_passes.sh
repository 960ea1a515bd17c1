#!/bin/bash
# Runs on the GPU box: the rocprofv3 counter passes behind profiles/ (one pass per run; block slot limits:
# 8 SQ, 4 TCC (FETCH_SIZE = 3, WRITE_SIZE = 2), 4 TCP, 2 TA, 2 TD, 2 GRBM).  Usage: tools/pmc_passes.sh OUTDIR [run_frame args]
set -u
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
PASSES=(
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 tools/run_frame.py "$@" > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i ($P) rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; exit $rc; fi
done
