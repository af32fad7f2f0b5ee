# FETCH_SIZE / WRITE_SIZE of the roofline kernel only (two --pmc passes) + its kernel-trace time: gpurun_out/attn_fetch.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_* gpurun_out/kt_f
export ATTN_DROPOUT=${ATTN_DROPOUT:-0.1}
for set_ in "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  tag=$(echo $set_ | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set_ --output-format csv -d gpurun_out/pmc_$tag -- python tools/prof_attn.py > gpurun_out/pmc_$tag.log 2>&1 || exit 1
done
python tools/pmc_summary.py gpurun_out qkv_attn4_kernel > gpurun_out/attn_fetch.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_f -- python tools/prof_attn.py > gpurun_out/kt_f.log 2>&1 || exit 1
grep qkv_attn4 $(ls -t gpurun_out/kt_f/*/*kernel_stats.csv | head -1) | cut -c1-180 >> gpurun_out/attn_fetch.txt
rm -rf gpurun_out/pmc_* gpurun_out/kt_f
cat gpurun_out/attn_fetch.txt
