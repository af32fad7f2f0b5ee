# attn_bwd5_kernel (five-product core) against attn_bwd_mfma_kernel (older core) at the bench size -> gpurun_out/attn_bwd5_stats.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/attn_bwd5_stats.txt
for cfg in "OLD= PDROP=0.1" "NODUMP=1 PDROP=0.1" "OLD= PDROP=0" "OLD= PDROP=0.1 S=101" "NODUMP=1 PDROP=0.1 S=101"; do
  rm -rf gpurun_out/kt_b
  env $cfg REPS=8 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_b -- python3 tools/prof_attn_bwd.py > gpurun_out/kt_b.log 2>&1 || exit 1
  f=$(ls -t gpurun_out/kt_b/*/*kernel_stats.csv | head -1)
  echo "== $cfg" >> gpurun_out/attn_bwd5_stats.txt
  python3 tools/kstats.py $f 8 >> gpurun_out/attn_bwd5_stats.txt
done
rm -rf gpurun_out/kt_b
cat gpurun_out/attn_bwd5_stats.txt
