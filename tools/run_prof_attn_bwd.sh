# attn_bwd_mfma_kernel under the ablation knobs of the tuning build -> gpurun_out/attn_bwd_ablation.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TUNING=1
: > gpurun_out/attn_bwd_ablation.txt
for d in 0 1 2 4 6; do
  rm -rf gpurun_out/kt_b
  MODCR_ATTN_BWD_DEBUG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_b -- python tools/prof_attn_bwd.py > gpurun_out/kt_b.log 2>&1 || exit 1
  echo "DEBUG=$d $(grep attn_bwd_mfma $(ls -t gpurun_out/kt_b/*/*kernel_stats.csv | head -1) | cut -d, -f2-4)" >> gpurun_out/attn_bwd_ablation.txt
done
rm -rf gpurun_out/kt_b
cat gpurun_out/attn_bwd_ablation.txt
