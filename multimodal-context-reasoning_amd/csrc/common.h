// Shared device/host helpers for libmodcr_hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/modcr_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MODCR_NEG (-10000.0f)   // the reference's additive mask value (modeling_transfomres.py:641)

// ---- host side error plumbing -------------------------------------------------------------
void modcr_set_error(const char* fmt, ...);
int modcr_check_launch(const char* what);

#define MODCR_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            modcr_set_error(__VA_ARGS__);        \
            return MODCR_ERR_INVALID;            \
        }                                        \
    } while (0)

static inline bool modcr_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- device helpers -----------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// exact erf GELU (transformers ACT2FN["gelu"]); tanh for the pooler / mapping networks
__device__ __forceinline__ float act_apply_exact(float v, int act) {
    if (act == MODCR_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (act == MODCR_ACT_TANH) return tanhf(v);
    return v;
}
// Branch-free forms for MFMA-kernel epilogues (bf16 outputs): erf by Abramowitz-Stegun 7.1.26
// (|err| <= 1.5e-7), tanh through exp.  No divergent control flow, so stores stay back to back.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float y = 1.0f - poly * t * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == MODCR_ACT_GELU) return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f));
    if (act == MODCR_ACT_TANH) return 1.0f - 2.0f * __frcp_rn(__expf(2.0f * v) + 1.0f);
    return v;
}

// Blocks b and b+8 share an XCD (round-robin dispatch, speed only).  Give each XCD a contiguous
// chunk of the logical tile order so neighbouring tiles hit the same L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
