cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -q --tb=short -m gpu -x -k "attn" > gpurun_out/attn_tests.log 2>&1; echo "tests exit $?"; tail -15 gpurun_out/attn_tests.log
grep -q passed gpurun_out/attn_tests.log && ! grep -q failed gpurun_out/attn_tests.log &&
MODCR_ATTN_DEBUG=8 timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -q --tb=short -m gpu -x -k "attn" > gpurun_out/attn_tests_exact.log 2>&1; echo "tests(exact) exit $?"; tail -3 gpurun_out/attn_tests_exact.log
timeout -k 10 300 python tools/ab_attn.py persist= nopersist=MODCR_ATTN_NOPERSIST=1 persistA=MODCR_ATTN_DEBUG=1 old=MODCR_ATTN_NO_V4=1 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-330
