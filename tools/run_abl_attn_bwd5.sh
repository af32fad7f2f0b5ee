# attn_bwd5_kernel under the timing-only ablation bits of the tuning build (results are wrong, only the clock counts)
#   1 = no block loads after the first two, 2 = no exponentials / dropout mask, 4 = no dQ phase, 8 = no dV / dK products,
#   16 = no barrier, 32 = no dQ stores
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TUNING=1
: > gpurun_out/attn_bwd5_ablation.txt
for d in ${ABL:-0 1 2 4 8 16 32 3 7 15 63}; do
  rm -rf gpurun_out/kt_b
  MODCR_ATTN_BWD_DEBUG=$d REPS=6 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_b -- python3 tools/prof_attn_bwd.py > gpurun_out/kt_b.log 2>&1 || exit 1
  echo "DEBUG=$d $(python3 tools/kstats.py $(ls -t gpurun_out/kt_b/*/*kernel_stats.csv | head -1) 12 | grep attn_bwd)" >> gpurun_out/attn_bwd5_ablation.txt
done
rm -rf gpurun_out/kt_b
cat gpurun_out/attn_bwd5_ablation.txt
