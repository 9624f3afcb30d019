#!/bin/bash
# PMC passes over one batch-sized GEMM shape: tools/pmc_small_gemm.sh M N K {nt|nn|tn}
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/pmc_small
mkdir -p $out
cd /tmp
export TMPDIR=/tmp
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/p$i -- python3 $repo/tools/gemm_one.py shape "$@" 5 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; continue; }
  python3 $repo/tools/pmc_summary.py $out/p$i gemm_kernel
done
