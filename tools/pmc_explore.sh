#!/bin/bash
# Several rocprofv3 --pmc passes over one LSB sort of 2^30 keys (SQ busy/wait, LDS conflicts, L2 / EA), each
# under its own timeout (a counter set the hardware cannot collect makes rocprofv3 abort and linger), then
# per-kernel means in gpurun_out/pmc_explore.txt.
run() { timeout -k 10 150 tools/pmc.sh "$1" "$2" 30 keys 1 || echo "pass $1 failed" >> gpurun_out/pmc_explore.txt; }
: > gpurun_out/pmc_explore.txt
run a1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
run a2 "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU"
run a4 "TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_HIT_sum"
run a5 "TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum"
run a6 "TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_CYCLE_sum"
run a7 "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
# (the four TA / TCP stall counters of the former pass a8 do not fit one pass: "Request exceeds the capabilities of the hardware")
run a8 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
run a9 "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
for t in a1 a2 a4 a5 a6 a7 a8 a9; do echo "== $t"; python tools/pmc_summary.py gpurun_out/pmc_$t; done >> gpurun_out/pmc_explore.txt
