# Round-5 PMC passes (each counter set its own rocprofv3 run, never combined with trace domains) + a kernel trace of the same command:
#   attn_n512        qkv_attn4_kernel<1,192,1,2,0> at the in-step size N = 512 (the roofline kernel, training mode, round-5 dropout mask)
#   attn_n256_train  the same at BASELINE config 2's N = 256;   attn_n256_eval  <1,192,0,2,0>, eval mode
#   attn_rob         qkv_attn4_kernel<1,128,1,2,1> at the RoBERTa body's shape N = 512, S = 106, H = 1024 with lse + dump
#   attn_bwd6        attn_bwd6_kernel<3,0,1,0> at N = 512, S = 180 (transposed reads on the r & 7 swizzle)
#   dw_ffn_up        dW = dY^T X of the FFN-up layer, [3072 x 768] over M = 92160: linear_bf16_p8_kernel<0,0,1,1,2,0> (half-TN, split-K)
# -> gpurun_out/r05_pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE")
one() {  # tag kernel-name-substring script  (environment of the script set by the caller)
  tag=$1; kern=$2; script=$3
  rm -rf gpurun_out/pmc_*
  for set in "${SETS[@]}"; do
    t=$(echo $set | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$t -- python3 $script > gpurun_out/pmc_$t.log 2>&1 || echo "pass $t failed"
  done
  python3 tools/pmc_summary.py gpurun_out $kern > gpurun_out/r05_pmc_$tag.txt 2>&1
  rm -rf gpurun_out/pmc_* gpurun_out/kt_g
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_g -- python3 $script > gpurun_out/kt_g.log 2>&1
  echo "# kernel-trace of the same command:" >> gpurun_out/r05_pmc_$tag.txt
  python3 tools/kstats.py $(ls -t gpurun_out/kt_g/*/*kernel_stats.csv | head -1) 5 >> gpurun_out/r05_pmc_$tag.txt
  rm -rf gpurun_out/kt_g
  echo "== $tag"; cat gpurun_out/r05_pmc_$tag.txt
}
for t in ${TAGS:-attn_n512 attn_n256_train attn_n256_eval attn_rob attn_bwd6 dw_ffn_up}; do
  case $t in
    attn_n512)       N=512 S=180 H=768 ITERS=6 ATTN_DROPOUT=0.1 one attn_n512 qkv_attn4_kernel tools/prof_attn.py ;;
    attn_n256_train) N=256 S=180 H=768 ITERS=6 ATTN_DROPOUT=0.1 one attn_n256_train qkv_attn4_kernel tools/prof_attn.py ;;
    attn_n256_eval)  N=256 S=180 H=768 ITERS=6 ATTN_DROPOUT=0 one attn_n256_eval qkv_attn4_kernel tools/prof_attn.py ;;
    attn_rob)        N=512 S=106 H=1024 ITERS=6 ATTN_DROPOUT=0.1 TRAINABLE=1 one attn_rob qkv_attn4_kernel tools/prof_attn.py ;;
    attn_bwd6)       N=512 S=180 REPS=3 one attn_bwd6 attn_bwd6 tools/prof_attn_bwd.py ;;
    dw_ffn_up)       M=92160 NN=3072 K=768 REPS=3 one dw_ffn_up "linear_bf16_p8_kernel<0, 0, 1, 1, 2" tools/prof_dw.py ;;
  esac
done
