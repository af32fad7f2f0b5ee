# PMC passes (separate, no trace domains) over the fused attention forward at the IN-STEP size N=512 (training-mode variant and
# the variant with the backward's side outputs is not part of this: bench.py's roofline kernel is qkv_attn4_kernel<1,192,1,2>)
# -> gpurun_out/r03_attn4_drop_pmc_n512.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_*
export N=512 ITERS=6 ATTN_DROPOUT=0.1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/prof_attn.py > gpurun_out/pmc_$tag.log 2>&1 || echo "pass $tag failed"
done
python3 tools/pmc_summary.py gpurun_out qkv_attn4_kernel > gpurun_out/r03_attn4_drop_pmc_n512.txt 2>&1
rm -rf gpurun_out/pmc_*
cat gpurun_out/r03_attn4_drop_pmc_n512.txt
