# PMC passes (separate, no trace domains) of the three encoder GEMM shapes at M = 92160 -> gpurun_out/r04_gemm_pmc_{ffn_up,ffn_down,proj}.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape() {  # tag M N K ACT OUT16
  tag=$1
  rm -rf gpurun_out/pmc_*
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
    t=$(echo $set | cut -d' ' -f1)
    M=$2 NN=$3 K=$4 ACT=$5 F16=$6 timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$t -- python3 tools/prof_gemm.py > gpurun_out/pmc_$t.log 2>&1 || echo "pass $t failed"
  done
  python3 tools/pmc_summary.py gpurun_out linear_bf16 > gpurun_out/r04_gemm_pmc_$tag.txt 2>&1
  rm -rf gpurun_out/pmc_*
  M=$2 NN=$3 K=$4 ACT=$5 F16=$6 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_g -- python3 tools/prof_gemm.py > gpurun_out/kt_g.log 2>&1
  echo "# kernel-trace of the same command:" >> gpurun_out/r04_gemm_pmc_$tag.txt
  python3 tools/kstats.py $(ls -t gpurun_out/kt_g/*/*kernel_stats.csv | head -1) 3 >> gpurun_out/r04_gemm_pmc_$tag.txt
  rm -rf gpurun_out/kt_g
  echo "== $tag"; cat gpurun_out/r04_gemm_pmc_$tag.txt
}
shape ffn_up 92160 3072 768 1 0
shape ffn_down 92160 768 3072 0 1
shape proj 92160 768 768 0 1
