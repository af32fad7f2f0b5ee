python -m pytest tests/test_hip_kernels.py -q --tb=short -m gpu -x 2>&1 | tail -15
python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids
MODCR_ATTN_OCC2=1 python tools/bench_kernels.py 2>&1 | grep qkv
