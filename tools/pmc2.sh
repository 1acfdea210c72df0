B="python3 bench.py --steps 1 --warmup 0 --profile-mode"
export HE355_DUAL_STREAM=0
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc2a -- $B > gpurun_out/pmc2a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/pmc2b -- $B > gpurun_out/pmc2b.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc2a | cut -c1-200
python3 tools/pmc_summary.py gpurun_out/pmc2b | cut -c1-200
