cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -q --tb=short -m gpu -x > gpurun_out/gpu_tests.log 2>&1; echo "tests exit $?" && tail -5 gpurun_out/gpu_tests.log &&
timeout -k 10 120 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_kernels.log; cat gpurun_out/bench_kernels.log &&
timeout -k 10 120 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_gemm.log; cat gpurun_out/bench_gemm.log &&
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_line.json 2> gpurun_out/bench_err.log; tail -1 gpurun_out/bench_line.json
