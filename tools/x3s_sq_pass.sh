cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MMR_LIB="$GRAFT_REPO_ROOT/multimodal-registration_amd/csrc/libmmr_hip_diag.so"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r04_x3s_SQ -- python3 tools/time_x3_layers.py --ab > gpurun_out/r04_x3s_SQ.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/r04_x3s_SQ2 -- python3 tools/time_x3_layers.py --ab > /dev/null 2>&1
echo done
