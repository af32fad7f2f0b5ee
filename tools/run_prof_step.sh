# step profile of bench.py under rocprofv3 (kernel trace + stats only) -> gpurun_out/r02_step_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-config3 "$@" > gpurun_out/prof_bench.log 2>&1 &&
f=$(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1) && cp $f gpurun_out/r02_step_kernel_stats.csv && rm -rf gpurun_out/prof_bench && tail -c 400 gpurun_out/prof_bench.log
