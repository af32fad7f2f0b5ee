# FFN-up shape: row-major tile walk (MODCR_GEMM_NGROUP=0) against column groups (6 / 3 / 4), tuning library; time + FETCH_SIZE
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export TUNING=1 M=${M:-92160} NN=${NN:-3072} K=${K:-768} ACT=${ACT:-1}
for g in ${NGROUPS:-0 6 3 4}; do
  rm -rf gpurun_out/kt_g gpurun_out/pmc_g
  MODCR_GEMM_NGROUP=$g timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_g -- python3 tools/prof_gemm.py > gpurun_out/kt_g.log 2>&1 || exit 1
  MODCR_GEMM_NGROUP=$g timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_g -- python3 tools/prof_gemm.py > gpurun_out/pmc_g.log 2>&1 || exit 1
  echo "NGROUP=$g $(python3 tools/kstats.py $(ls -t gpurun_out/kt_g/*/*kernel_stats.csv | head -1) 3 | grep linear_bf16 | cut -c1-40) $(python3 tools/pmc_summary.py gpurun_out linear_bf16 | grep FETCH)"
done
rm -rf gpurun_out/kt_g gpurun_out/pmc_g
