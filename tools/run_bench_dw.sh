# dW forms at config 3's row count, then per-kernel times of both forms under rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
M=92160 timeout -k 10 300 python3 tools/bench_dw.py > gpurun_out/bench_dw.log 2>&1 || exit 1
cat gpurun_out/bench_dw.log
rm -rf gpurun_out/prof_dw
M=92160 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dw -- python3 tools/bench_dw.py > /dev/null 2> gpurun_out/prof_dw.err || exit 1
f=$(ls gpurun_out/prof_dw/*/*kernel_stats.csv | head -1) && cp $f gpurun_out/dw_kernel_stats.csv && rm -rf gpurun_out/prof_dw
python3 tools/kstats.py gpurun_out/dw_kernel_stats.csv 12
