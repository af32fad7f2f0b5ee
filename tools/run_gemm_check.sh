cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_hip_kernels.py -q --tb=short -m gpu -x -k "linear" > gpurun_out/gemm_tests.log 2>&1; echo "tests exit $?"; tail -6 gpurun_out/gemm_tests.log
grep -q passed gpurun_out/gemm_tests.log && ! grep -q failed gpurun_out/gemm_tests.log &&
for d in 1 0 1 0; do echo "direct=$d"; MODCR_GEMM_DIRECT=$d timeout -k 10 100 python tools/bench_kernels.py 2>&1 | grep -A1 "proj\|ffn_down"; done
