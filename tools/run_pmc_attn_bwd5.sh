# PMC passes (separate, no trace domains) over attn_bwd5_kernel at the bench size -> gpurun_out/attn_bwd5_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_*
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  REPS=3 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/prof_attn_bwd.py > gpurun_out/pmc_$tag.log 2>&1 || echo "pass $tag failed"
done
python3 tools/pmc_summary.py gpurun_out attn_bwd5 > gpurun_out/attn_bwd5_pmc.txt 2>&1
rm -rf gpurun_out/pmc_*
cat gpurun_out/attn_bwd5_pmc.txt
