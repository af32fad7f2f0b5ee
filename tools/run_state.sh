cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -q --tb=short -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "tests exit $?" && tail -8 gpurun_out/gpu_tests.log &&
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log; tail -1 gpurun_out/bench_line.json | cut -c1-250; tail -1 gpurun_out/bench_line.json | grep -o '"roofline.*'
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dropout 0 2> gpurun_out/bench_err2.log | cut -c1-250
bash tools/run3.sh 2>&1 | sed -n 1,8p
