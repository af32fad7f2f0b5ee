cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_hip_kernels.py -q --tb=short -m gpu -x -k "linear" > gpurun_out/gemm_tests.log 2>&1; echo "tests exit $?"; tail -12 gpurun_out/gemm_tests.log
grep -q passed gpurun_out/gemm_tests.log && ! grep -q failed gpurun_out/gemm_tests.log &&
SHAPES=46080x3072x768,46080x768x3072,46080x768x768,25856x3072x768,25856x768x3072 timeout -k 10 120 python tools/bench_gemm.py 2>&1 | grep linear &&
MODCR_GEMM_T192=0 SHAPES=46080x3072x768,46080x768x3072,46080x768x768,25856x3072x768,25856x768x3072 timeout -k 10 120 python tools/bench_gemm.py 2>&1 | grep linear &&
timeout -k 10 120 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids
