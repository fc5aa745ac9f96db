mkdir -p gpurun_out/r02d
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
S2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"
S3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
S4="TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
S5="TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
rocprofv3 -L > gpurun_out/r02d/counters_list.txt 2>&1
python scripts/pmc_sets.py --scene 2 --env BRT_FORCE_GLOBAL_SCENE=1 -- "$S1" "$S2" "$S3" "$S4" "$S5" > gpurun_out/r02d/cfg5_mode0.json 2>&1
python scripts/pmc_sets.py --scene 2 -- "$S1" "$S2" "$S3" "$S4" "$S5" > gpurun_out/r02d/cfg5_mode2.json 2>&1
cat gpurun_out/r02d/cfg5_mode0.json gpurun_out/r02d/cfg5_mode2.json
