# Round profile set (run on the GPU box: gpurun -- 'bash tools/run_prof_all.sh'):
#   1. rocprofv3 --kernel-trace --stats of bench.py (the step profile; also prints the bench line of the same run)
#   2. one --pmc set per pass (never combined with trace domains) over tools/prof_attn.py for the training-mode
#      (attention dropout 0.1) and the eval-mode variant of the roofline kernel + their kernel-trace stats
# Results land in gpurun_out/; copy the summaries into profiles/ as r<NN>_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench gpurun_out/pmc_* gpurun_out/kt_*
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 > gpurun_out/prof_bench.log 2>&1 &&
f=$(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) && cp $f gpurun_out/bench_kernel_stats.csv && rm -rf gpurun_out/prof_bench &&
grep -o '{"metric.*' gpurun_out/prof_bench.log > gpurun_out/bench_line_under_rocprof.json
for variant in "0.1 train" "0 eval"; do
  set -- $variant
  export ATTN_DROPOUT=$1
  rm -rf gpurun_out/pmc_*
  for set_ in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
    tag=$(echo $set_ | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 --pmc $set_ --output-format csv -d gpurun_out/pmc_$tag -- python tools/prof_attn.py > gpurun_out/pmc_$tag.log 2>&1 || exit 1
  done
  python tools/pmc_summary.py gpurun_out qkv_attn4_kernel > gpurun_out/attn_pmc_$2.txt 2>&1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$2 -- python tools/prof_attn.py > gpurun_out/kt_$2.log 2>&1 || exit 1
  cp $(ls -t gpurun_out/kt_$2/*/*kernel_stats.csv | head -1) gpurun_out/attn_kernel_stats_$2.csv
done
rm -rf gpurun_out/pmc_* gpurun_out/kt_*/
cat gpurun_out/attn_pmc_train.txt | head -30
head -12 gpurun_out/bench_kernel_stats.csv | cut -c1-200
