#!/usr/bin/env python
"""GEMM-side breakdown of a config-3 step profile (rocprofv3 kernel_stats.csv of `bench.py --train-encoders`, 13 steps):
    python tools/c3_gemm_breakdown.py profiles/r03_c3_step_kernel_stats.csv > profiles/r03_c3_gemm_breakdown.txt"""
import csv
import sys

ROLES = [
    ("linear_bf16_p8_kernel<0, 0, 1, 1, 0>", "dW = dY^T X with both operands transposed (the heads' products and shapes the half-TN form does not take), split-K, fp32 partials"),
    ("linear_bf16_p8_kernel<0, 0, 1, 1, 2>", "dW, half-TN form: the wide operand token-major through transposed LDS reads (4 per trainable layer; two of them formed transposed)"),
    ("linear_bf16_t192_kernel<0, 0, 1, 1>", "trainable forward: proj / FFN-down with fp32 rows for the LayerNorm pass (2 per layer)"),
    ("linear_bf16_p8_kernel<5, 0, 0, 1, 0>", "trainable forward: FFN-up, writes gelu(u) and u (kept GELU input)"),
    ("linear_bf16_p8_kernel<4, 1, 0, 0, 0>", "backward: FFN-down dX x gelu'(u) -> d_u (bf16) + its column sums (bias gradient)"),
    ("linear_bf16_t192_kernel<0, 2, 1, 0>", "backward: FFN-up dX + residual-branch gradient (fp32 in, fp32 out)"),
    ("linear_bf16_t192_kernel<0, 2, 0, 0>", "backward: QKV dX + residual-branch gradient -> dx (bf16)"),
    ("linear_bf16_t192_kernel<0, 0, 0, 1>", "backward: proj dX -> d_ctx (bf16); frozen-pass GEMMs with bf16 out"),
    ("linear_bf16_p8_kernel<0, 0, 2, 1, 0>", "image-only frozen pass: proj / FFN-down with IEEE-half rows"),
    ("linear_bf16_p8_kernel<1, 0, 0, 1, 0>", "image-only frozen pass: FFN-up + GELU"),
    ("transpose256_kernel", "operand transposes of the dW products (the narrow operand: 4 x [M, H] per trainable layer)"),
    ("reduce_partials_transposed_kernel", "split-K reduction of the dW products formed transposed"),
    ("reduce_partials_kernel", "split-K reduction of the dW partials"),
    ("colsum_bf16_block_kernel", "bias gradient of the QKV projection (column sums of dqkv)"),
    ("transpose64_kernel", "W^T for the dX products / small operands"),
]
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
print("config 3 (both encoders trained, 128 examples, M = 92160 rows per encoder pass) -- GEMM-side breakdown of %s" % sys.argv[1])
print("(rocprofv3 --kernel-trace --stats over %d steps; per-step = total / %d).  FLOP: the four dW products of a layer 1.305 TFLOP;" % (steps, steps))
print("FFN-up / FFN-down forward or dX 0.435 TFLOP each; proj 0.109; QKV dX 0.326.\n")
print("%8s %10s %9s  %s" % ("calls/st", "avg us", "ms/step", "kernel : role"))
tot = 0.0
for r in rows:
    name = r["Name"]
    role = next((v for k, v in ROLES if k in name), None)
    if role is None:
        continue
    ms = float(r["TotalDurationNs"]) / steps / 1e6
    tot += ms
    short = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("%8.1f %10.1f %9.3f  %s : %s" % (int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, ms, short, role))
print("\nsum of the rows above: %.1f ms per step" % tot)
