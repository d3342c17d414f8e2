cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MMR_LIB="$GRAFT_REPO_ROOT/multimodal-registration_amd/csrc/libmmr_hip.so"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r04b_pmc_ncc_SQ -- python3 bench.py --workload ncc --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r04b_pmc_ncc_SQ.err || exit 1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/r04b_pmc_ncc_SQ2 -- python3 bench.py --workload ncc --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/r04b_pmc_ncc_SQ2.err
echo done
