cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -q --tb=short -m gpu -x -k "attn" > gpurun_out/attn_tests.log 2>&1; echo "tests exit $?"; tail -15 gpurun_out/attn_tests.log
grep -q passed gpurun_out/attn_tests.log && ! grep -q failed gpurun_out/attn_tests.log &&
MODCR_ATTN_DEBUG=8 timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -q --tb=short -m gpu -x -k "attn" > gpurun_out/attn_tests_exact.log 2>&1; echo "tests(exact) exit $?"; tail -3 gpurun_out/attn_tests_exact.log
timeout -k 10 120 python tools/bench_kernels.py 2>&1 | grep qkv &&
MODCR_ATTN_DEBUG=1 timeout -k 10 120 python tools/bench_kernels.py 2>&1 | grep qkv &&
MODCR_ATTN_DEBUG=8 timeout -k 10 120 python tools/bench_kernels.py 2>&1 | grep qkv
