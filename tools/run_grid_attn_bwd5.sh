cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TUNING=1
for gsz in ${GRIDS:-256 512 768 1024}; do
  rm -rf gpurun_out/kt_b
  MODCR_ATTN_BWD_GRID=$gsz REPS=4 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_b -- python3 tools/prof_attn_bwd.py > gpurun_out/kt_b.log 2>&1 || exit 1
  echo "GRID=$gsz $(python3 tools/kstats.py $(ls -t gpurun_out/kt_b/*/*kernel_stats.csv | head -1) 12 | grep attn_bwd5)"; grep -m1 "occupancy" gpurun_out/kt_b.log
done
