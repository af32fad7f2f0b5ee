cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log
tail -1 gpurun_out/bench_line.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
f=$(ls gpurun_out/prof_bench/*/*kernel_stats.csv | head -1); cp $f gpurun_out/bench_kernel_stats.csv; head -40 $f
