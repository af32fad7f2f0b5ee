cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -q --tb=short -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "tests exit $?" && tail -6 gpurun_out/gpu_tests.log &&
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log; tail -1 gpurun_out/bench_line.json | cut -c1-250; tail -1 gpurun_out/bench_line.json | grep -o '"roofline.*'
MODCR_ATTN_NO_V4S=1 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/bench_err2.log | cut -c1-250
timeout -k 10 500 python bench.py --steps 3 --warmup 1 --with-roberta 2>&1 | tail -1 | cut -c1-220
