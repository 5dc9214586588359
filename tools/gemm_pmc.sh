# SQ counter passes over tools/gemm_pmc_probe.py (kernel trace only, one counter set per run)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace -d $O/gpmc_$i -o run -- python3 tools/gemm_pmc_probe.py > $O/gpmc_$i.log 2>&1
  python3 tools/pmc_by_grid.py $O/gpmc_$i/run_results.db gemm_bf16 >> $O/r02_gemm_pmc.txt
  rm -rf $O/gpmc_$i
done
cat $O/r02_gemm_pmc.txt
