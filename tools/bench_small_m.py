"""M = 256 GEMMs of the trainable heads (mapping networks: 768 -> 3840 -> 5120 with the 3-term bf16 split, K tripled)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for m, n, k in ((256, 3840, 768), (256, 5120, 3840), (256, 3840, 5120), (256, 768, 3840), (256, 768, 1536), (256, 768, 768)):
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.02; b = torch.randn(n, device=dev)
    xs, ws = mh.split3(x, 0), mh.split3(w, 1)
    us = t(lambda: mh.linear(xs, ws, b, act=mh.ACT_TANH, out_dtype=mh.F32))
    print("M=%d N=%d K=%d (x3 = %d): %.1f us  %.0f TF (bf16 flops incl. the split)" % (m, n, k, 3 * k, us, 2.0 * m * n * 3 * k / us / 1e6), flush=True)
