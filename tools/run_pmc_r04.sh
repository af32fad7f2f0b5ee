# Round-4 PMC passes (each counter set its own rocprofv3 run, never combined with trace domains) + a kernel trace of the same command:
#   ffn_up      linear_bf16_p8_kernel<1,0,0,1,0,1> (FFN-up, GELU, M = 92160: the FINAL kernel, seamless ring)
#   attn_n512   qkv_attn4_kernel<1,192,1,2> at the in-step size N = 512 (the roofline kernel)
#   attn_rob    qkv_attn4_kernel<1,128,1,2> at the RoBERTa body's shape N = 512, S = 106, H = 1024 with lse + dump
#   attn_bwd6   attn_bwd6_kernel<3,0,1,0> at N = 512, S = 180
# -> gpurun_out/r04_pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE")
one() {  # tag kernel-name-substring script  (environment of the script set by the caller)
  tag=$1; kern=$2; script=$3
  rm -rf gpurun_out/pmc_*
  for set in "${SETS[@]}"; do
    t=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$t -- python3 $script > gpurun_out/pmc_$t.log 2>&1 || echo "pass $t failed"
  done
  python3 tools/pmc_summary.py gpurun_out $kern > gpurun_out/r04_pmc_$tag.txt 2>&1
  rm -rf gpurun_out/pmc_* gpurun_out/kt_g
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_g -- python3 $script > gpurun_out/kt_g.log 2>&1
  echo "# kernel-trace of the same command:" >> gpurun_out/r04_pmc_$tag.txt
  python3 tools/kstats.py $(ls -t gpurun_out/kt_g/*/*kernel_stats.csv | head -1) 4 >> gpurun_out/r04_pmc_$tag.txt
  rm -rf gpurun_out/kt_g
  echo "== $tag"; cat gpurun_out/r04_pmc_$tag.txt
}
for t in ${TAGS:-ffn_up attn_n512 attn_rob attn_bwd6}; do
  case $t in
    ffn_up)    M=92160 NN=3072 K=768 ACT=1 F16=0 one ffn_up linear_bf16 tools/prof_gemm.py ;;
    attn_n512) N=512 S=180 H=768 ITERS=6 ATTN_DROPOUT=0.1 one attn_n512 qkv_attn4_kernel tools/prof_attn.py ;;
    attn_rob)  N=512 S=106 H=1024 ITERS=6 ATTN_DROPOUT=0.1 TRAINABLE=1 one attn_rob qkv_attn4_kernel tools/prof_attn.py ;;
    attn_bwd6) N=512 S=180 REPS=3 one attn_bwd6 attn_bwd6 tools/prof_attn_bwd.py ;;
  esac
done
