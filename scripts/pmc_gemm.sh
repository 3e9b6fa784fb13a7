#!/bin/bash
# Counter passes over the split GEMMs alone (scripts/bench_gemm.py, ONLY=SPLIT: the four encoder shapes, forward,
# dgrad and both wgrad forms), each pass in its own run with no trace flags, program directly after `--`:
#   gpurun --timeout 900 -- 'bash scripts/pmc_gemm.sh r03'
# then here:  python scripts/summarize_pmc_gemm.py r03
# Pass 1: LDS / issue-stall counters.  Pass 2: wave cycles, MFMA busy, clock.  Pass 3: L2 hit / miss / requests.
set -e
TAG=${1:?tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export ONLY=${ONLY:-SPLIT} ROUNDS=${ROUNDS:-3}
B="python3 $R/scripts/bench_gemm.py"
O=$R/gpurun_out/pmc_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_trace -- $B > ${O}_trace.log 2>&1
echo "trace done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d ${O}_lds -- $B > ${O}_lds.log 2>&1
echo "lds done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d ${O}_sq -- $B > ${O}_sq.log 2>&1
echo "sq done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d ${O}_tcc -- $B > ${O}_tcc.log 2>&1
echo "tcc done"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d ${O}_tcp -- $B > ${O}_tcp.log 2>&1 || echo "tcp pass failed (counter names)"
echo "tcp done"
tail -12 ${O}_trace.log
