# scratch batch for one gpurun call (edited per experiment): TA / TCP / LDS counters of the pool scene (cited in DESIGN section 5)
export TMPDIR=/tmp
bash tools/pmc_probe.sh pool backrooms_pool 32 "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" > gpurun_out/probe_pool.log 2>&1
cat gpurun_out/probe_pool.log
