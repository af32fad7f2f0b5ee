# kernel time + WRITE_SIZE of the roofline kernel's eval-mode and training-mode variants, same box, same session
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 0.1; do
  ATTN_DROPOUT=$v timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/av_pmc_$v -- python tools/prof_attn.py > gpurun_out/av_pmc_$v.log 2>&1 || exit 1
  ATTN_DROPOUT=$v ITERS=20 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/av_kt_$v -- python tools/prof_attn.py > gpurun_out/av_kt_$v.log 2>&1 || exit 1
done
