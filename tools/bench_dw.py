"""dW = dY^T X at the encoder-backward shapes: the half-TN form (dY^T transposed, X token-major: the default), both operands transposed, and the full TN form (both token-major)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh
mh.use_tuning_library(True)      # MODCR_GEMM_TN exists in the tuning build only
dev = torch.device("cuda")
MS = int(os.environ.get("M", 46080))
for m, n, k in ((MS, 768, 768), (MS, 3072, 768), (MS, 768, 3072), (MS, 2304, 768), (46080, 768, 768), (46080, 3072, 768), (46080, 768, 3072), (46080, 2304, 768), (27136, 1024, 1024), (27136, 4096, 1024), (27136, 1024, 4096)):
    dy = torch.randn(m, n, device=dev).to(torch.bfloat16)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    dw, db = torch.empty(n, k, device=dev), torch.empty(n, device=dev)
    res = []
    for env in ({"MODCR_GEMM_TN": "0", "MODCR_GEMM_HALF_TN": "2"}, {"MODCR_GEMM_TN": "0", "MODCR_GEMM_HALF_TN": "0"}, {"MODCR_GEMM_TN": "1", "MODCR_GEMM_HALF_TN": "0"}):
        os.environ.update(env)
        for _ in range(3):
            mh.linear_bwd_weight(dy, x, dw, db, mfma=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            mh.linear_bwd_weight(dy, x, dw, db, mfma=True)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    if n > k:
        os.environ.update({"MODCR_GEMM_TN": "0", "MODCR_GEMM_HALF_TN": "1"})
        for _ in range(3):
            mh.linear_bwd_weight(dy, x, dw, None, mfma=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            mh.linear_bwd_weight(dy, x, dw, None, mfma=True)
        e1.record(); torch.cuda.synchronize()
        print("   the same without db (product formed transposed, dY token-major): %.1f us" % (e0.elapsed_time(e1) / 10 * 1e3), flush=True)
    print("M=%d N=%d K=%d: half-TN (X token-major) %.1f us (%.0f TF incl. transposes + reduce + db), both transposed %.1f us, full TN %.1f us" % (
        m, n, k, res[0], 2.0 * m * n * k / res[0] / 1e6, res[1], res[2]), flush=True)
