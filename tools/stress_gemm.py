"""Stress the persistent GEMM kernels at the backward dX shapes (no bias: nothing serialises the next tile's DMAs against the
epilogue's stores), cache flushed before each launch; every launch is compared with a torch fp32 product.  Found the
store / LDS-DMA retirement-order bug of round 1 (1 bad launch in ~150 before the fix, 0 of 4 x 500 launches after)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh
torch.manual_seed(0)
dev = torch.device("cuda")
junk1 = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
junk2 = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
def stress(name, m, n, k, out_dtype, bias, iters):
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.03).to(torch.bfloat16)
    b = torch.randn(n, device=dev) if bias else None
    ref = (a.float() @ w.float().t()) + (b if bias else 0)
    bad = 0
    for i in range(iters):
        junk1.copy_(junk2)
        out = mh.linear(a, w, b, out_dtype=out_dtype).float()
        d = (out - ref).abs()
        if not float(d.max()) < 0.5:
            bad += 1
            rows = torch.nonzero(~(d.max(1).values < 0.5)).flatten()
            print("  ", name, "iter", i, "bad rows", rows.numel(), "first", int(rows[0]), "last", int(rows[-1]), "rows%192 first", int(rows[0]) % 192, flush=True)
    print(name, "bad launches:", bad, "of", iters, flush=True)
it = int(os.environ.get("ITERS", "150"))
stress("t192 K=2304 bf16 nobias", 46080, 768, 2304, mh.BF16, False, it)
stress("t192 K=2304 bf16 bias", 46080, 768, 2304, mh.BF16, True, it)
stress("t192 K=768 f32 nobias", 46080, 768, 768, mh.F32, False, it)
stress("p8 N=3072 K=768 bf16 nobias", 46080, 3072, 768, mh.BF16, False, it)
# round 4: the seamless-ring form (whole tiles, 16-bit output): the ring carries the next tile's half-tiles through the epilogue's stores
stress("p8 seamless N=3072 K=768 bf16 bias M=92160 (17 tiles per workgroup)", 92160, 3072, 768, mh.BF16, True, it)
stress("p8 seamless N=768 K=768 f16 nobias M=51712", 51712, 768, 768, mh.F16, False, it)
stress("p8 seamless N=4096 K=1024 bf16 nobias M=23552", 23552, 4096, 1024, mh.BF16, False, it)
# round 6: activation operands larger than the Infinity Cache walk their tiles from the last to the first (rev_walk_for, csrc/gemm.hip):
# FFN-down on both persistent kernels (t192 at M = 92160, the seamless-ring p8 at the batched global_enc passes' M = 143872)
stress("t192 reversed walk N=768 K=3072 f16 bias M=92160", 92160, 768, 3072, mh.F16, True, it)
stress("p8 seamless reversed walk N=768 K=3072 f16 bias M=143872", 143872, 768, 3072, mh.F16, True, max(it // 2, 1))


def stress_dw(name, m, n, k, with_db, iters):
    """weight-gradient products (half-TN forms of the 256 x 256 kernel): every launch bit-equal to the first, the first against a
    float64 product on a column sample"""
    dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    first, bad = None, 0
    for i in range(iters):
        junk1.copy_(junk2)
        dw = torch.empty(n, k, device=dev)
        db = torch.empty(n, device=dev) if with_db else None
        mh.linear_bwd_weight(dy, x, dw, db, mfma=True)
        if first is None:
            cols = slice(0, k, 97)
            ref = (dy.double().t() @ x[:, cols].double()).float()
            err = float((dw[:, cols] - ref).abs().max() / ref.abs().max())
            assert err < 2e-3, (name, err)
            first = dw.clone()
        elif not torch.equal(dw, first):
            bad += 1
            print("  ", name, "iter", i, "differs in", int((dw != first).sum()), "elements", flush=True)
    print(name, "bad launches:", bad, "of", iters, flush=True)


stress_dw("dW [768 x 3072] half-TN (X token-major)", 92160, 768, 3072, True, it)
stress_dw("dW [768 x 768] half-TN", 92160, 768, 768, True, it)
stress_dw("dW [3072 x 768] formed transposed, no db", 92160, 3072, 768, False, it)
stress_dw("dW [2304 x 768] formed transposed + column sums", 92160, 2304, 768, True, it)
stress_dw("dW [4096 x 1024] formed transposed, M = 54272", 54272, 4096, 1024, False, it)
