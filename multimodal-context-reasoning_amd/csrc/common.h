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
// dX = dY.W + res (fp32 [M,K] or NULL) -- gemm.hip; library-internal (hidden), used by the attention backward composite
__attribute__((visibility("hidden"))) int modcr_linear_bwd_input_res(const void* dY, int64_t lddy, int32_t dy_dtype, const void* W, int64_t ldw,
                                                                      const float* res, int64_t ldr, void* dX, int64_t lddx, int32_t M, int32_t N,
                                                                      int32_t K, int32_t dtype, int32_t out_dtype, void* workspace,
                                                                      int64_t workspace_bytes, modcr_stream_t stream);

#define MODCR_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            modcr_set_error(__VA_ARGS__);        \
            return MODCR_ERR_INVALID;            \
        }                                        \
    } while (0)

static inline bool modcr_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Launch-side caches (the "LDS attribute already set" flags, the CU count) are properties of the CURRENT device: a process that
// touches a second GPU (tests, tools) must not launch a 130 KB-LDS kernel unconfigured there or size a persistent grid from the
// first device.  One slot per device ordinal; ordinals past the table share the last slot's flag only for the CU count default.
#define MODCR_MAX_DEV 32
static inline int modcr_device_index() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < MODCR_MAX_DEV ? dev : MODCR_MAX_DEV - 1;
}
static inline int modcr_device_cus() {
    static int cus[MODCR_MAX_DEV] = {};
    const int d = modcr_device_index();
    if (!cus[d]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v < 1) v = 256;
        cus[d] = v;
    }
    return cus[d];
}

// Tuning / debug knobs (A/B runs of kernel variants, timing-only ablations that skip work) exist ONLY in
// libmodcr_hip_tuning.so (`make tuning`, -DMODCR_TUNING; used by tools/ and by the tests that force a code path).  The
// product library reads NO environment variable: every knob is its compile-time default there and the timing-only
// branches (MODCR_DBG) are compiled out.  In the tuning build a knob is re-read on every call.
#ifdef MODCR_TUNING
#include <stdlib.h>
static inline int modcr_knob_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static inline int modcr_knob_set(const char* name) { return getenv(name) ? 1 : 0; }
#define MODCR_DBG(expr) (expr)
#else
#define modcr_knob_int(name, dflt) (dflt)
#define modcr_knob_set(name) 0
#define MODCR_DBG(expr) 0
#endif

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
// Branch-free forms for MFMA-kernel epilogues (bf16-path outputs).  The FFN-up epilogue applies
// GELU to 65536 values per 256x256 tile: measured ~100 us of a 415 us launch with an
// Abramowitz-Stegun erf, so the math runs two values per instruction (v_pk_fma_f32 /
// v_pk_mul_f32) and uses the cheapest form whose error stays under the bf16 rounding of the output.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 gelu2(f32x2 v) {
    // x * Phi(x) with Phi(x) = 1 / (1 + exp(-(1.59576912 x + 0.07135481 x^3)))  (Page's logistic form
    // of the normal CDF, |Phi err| <= 1.4e-4; identical to the "tanh" GELU).  |gelu err| <= 5e-4 at
    // |x| ~ 2-3, below half a bf16 ulp of the result there; the fp32 parity path uses erff.
    // 5 packed ops + 2 transcendentals per pair instead of 12 + 4.
    const f32x2 x2 = v * v;
    const f32x2 u = v * (x2 * (-0.07135481283f * 1.44269504089f) + (-1.59576912161f * 1.44269504089f));
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(u.x); e.y = __builtin_amdgcn_exp2f(u.y);
    const f32x2 d = e + 1.0f;
    f32x2 rinv;
    rinv.x = __builtin_amdgcn_rcpf(d.x); rinv.y = __builtin_amdgcn_rcpf(d.y);   // v_rcp_f32 (1 ulp); __frcp_rn expands to the 10-instruction IEEE division
    return v * rinv;
}
// (A transcendental-free form -- erf(t / sqrt 2) ~ t P(t^2), odd degree-13 minimax on |t| <= 4, |gelu err| 1.9e-4, tools/fit_gelu.py:
// 11 packed ops + 2 v_med3 per pair -- was built twice and measured slower both times (round 4: FFN-up 414 vs 409 us, epilogue
// 13.0 k vs 11.4 k cycles of the slower wave): on gfx950 a packed fp32 op costs about what a v_exp_f32 / v_rcp_f32 does.)
// d/dx of the exact GELU, Phi(x) + x phi(x), with the same logistic Phi as gelu2 (bf16-path epilogues only)
#define MODCR_ACT_GELU_GRAD 3      // internal epilogue code: out = gelu'(acc + bias) * residual operand (FFN-up backward)
#define MODCR_ACT_MUL_GELU_GRAD 4  // internal epilogue code: out = acc * gelu'(residual operand) (FFN-down dX with the saved GELU input)
#define MODCR_ACT_GELU_KEEP 5      // internal epilogue code: out = gelu(acc + bias), and acc + bias goes to LinearArgs::C2 (trainable FFN-up)
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 v) {
    const f32x2 x2 = v * v;
    const f32x2 u = v * (x2 * (-0.07135481283f * 1.44269504089f) + (-1.59576912161f * 1.44269504089f));
    const f32x2 w = x2 * (-0.5f * 1.44269504089f);
    f32x2 e, g;
    e.x = __builtin_amdgcn_exp2f(u.x); e.y = __builtin_amdgcn_exp2f(u.y);
    g.x = __builtin_amdgcn_exp2f(w.x); g.y = __builtin_amdgcn_exp2f(w.y);
    const f32x2 d = e + 1.0f;
    f32x2 rinv;
    rinv.x = __builtin_amdgcn_rcpf(d.x); rinv.y = __builtin_amdgcn_rcpf(d.y);
    return rinv + v * (g * 0.39894228040143267794f);
}
__device__ __forceinline__ f32x2 tanh2(f32x2 v) {
    const f32x2 a = v * 2.88539008177792681472f;                    // exp(2v) = exp2(2 v log2 e)
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(a.x); e.y = __builtin_amdgcn_exp2f(a.y);
    const f32x2 d = e + 1.0f;
    f32x2 rinv;
    rinv.x = __builtin_amdgcn_rcpf(d.x); rinv.y = __builtin_amdgcn_rcpf(d.y);   // v_rcp_f32 (1 ulp); __frcp_rn expands to the 10-instruction IEEE division
    return 1.0f - rinv * 2.0f;
}
// v[0..3] = act(v[0..3] + b[0..3])
__device__ __forceinline__ void bias_act4(float (&v)[4], const float (&b)[4], int act) {
    f32x2 lo = {v[0] + b[0], v[1] + b[1]}, hi = {v[2] + b[2], v[3] + b[3]};
    if (act == MODCR_ACT_GELU || act == MODCR_ACT_GELU_KEEP) { lo = gelu2(lo); hi = gelu2(hi); }
    if (act == MODCR_ACT_TANH) { lo = tanh2(lo); hi = tanh2(hi); }
    if (act == MODCR_ACT_GELU_GRAD) { lo = gelu_grad2(lo); hi = gelu_grad2(hi); }
    v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
}
// how an epilogue combines its value with the residual operand: added, (GELU_GRAD) multiplied, or (MUL_GELU_GRAD) multiplied
// by gelu'(operand)
template <int ACT> __device__ __forceinline__ float res_apply(float v, float r) { return ACT == MODCR_ACT_GELU_GRAD ? v * r : v + r; }
template <int ACT> __device__ __forceinline__ void res_apply4(float (&v)[4], const float (&r)[4]) {
    if constexpr (ACT == MODCR_ACT_MUL_GELU_GRAD) {
        const f32x2 g0 = gelu_grad2(f32x2{r[0], r[1]}), g1 = gelu_grad2(f32x2{r[2], r[3]});
        v[0] *= g0.x; v[1] *= g0.y; v[2] *= g1.x; v[3] *= g1.y;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = res_apply<ACT>(v[e], r[e]);
    }
}
__device__ __forceinline__ float act_apply(float v, int act) {      // scalar tail path
    f32x2 t = {v, v};
    if (act == MODCR_ACT_GELU) t = gelu2(t);
    if (act == MODCR_ACT_TANH) t = tanh2(t);
    return t.x;
}

// counter-based dropout decisions (see modcr_dropout below): the decision of counter c is field (c & 3) of the two hash
// words of its GROUP c >> 2 -- four 15-bit uniforms per hash (bits 0-14 and 16-30 of x and y), kept iff uniform >= the
// 15-bit threshold round(p * 2^15).  One full 32-bit finaliser + one multiply-xorshift per four consecutive counters: the
// row kernels, whose lanes own four consecutive columns, hash once per 16-byte piece (the per-element two-finaliser hash
// of round 1 made the LayerNorm + dropout pass VALU-bound: six v_mul_lo_u32 per element).
__device__ __forceinline__ void drop_words(uint64_t seed, uint64_t grp, uint32_t& x, uint32_t& y) {
    x = (uint32_t)grp * 0x9E3779B1u ^ (uint32_t)(grp >> 32) * 0x85EBCA77u ^ (uint32_t)seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    y = x * 0x2C1B3C6Du + (uint32_t)(seed >> 32);
    y ^= y >> 15;
}
__device__ __forceinline__ uint32_t drop_field(uint32_t x, uint32_t y, int f) {
    return (((f & 2) ? y : x) >> ((f & 1) * 16)) & 0x7fffu;
}
__device__ __forceinline__ float drop_apply(float v, uint64_t seed, uint64_t ctr, uint32_t thr, float scale) {
    uint32_t x, y;
    drop_words(seed, ctr >> 2, x, y);
    return drop_field(x, y, (int)(ctr & 3)) >= thr ? v * scale : 0.f;
}
// four consecutive counters ctr .. ctr + 3: one hash when they share a group (ctr % 4 == 0, the row kernels' case)
__device__ __forceinline__ void drop_apply4(float (&v)[4], uint64_t seed, uint64_t ctr, uint32_t thr, float scale) {
    if ((ctr & 3) == 0) {
        uint32_t x, y;
        drop_words(seed, ctr >> 2, x, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = drop_field(x, y, j) >= thr ? v[j] * scale : 0.f;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = drop_apply(v[j], seed, ctr + (uint64_t)j, thr, scale);
    }
}

// One LDS-DMA instruction in its scalar-base form: 16 bytes per lane from (uniform 64-bit base in SGPRs +
// 32-bit per-lane byte offset) to LDS at (wave-uniform address in M0) + 16 * lane.  Written as asm because
// inside a loop hipcc turns base + offset into per-lane 64-bit pointers (two VGPRs and a v_lshl_add_u64
// per source, spilled under the register pressure of the persistent tile loop).
__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, const void* lds_dst) {
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(la) : "memory", "m0");
}

// Blocks b and b+8 share an XCD (round-robin dispatch, speed only).  Give each XCD a contiguous
// chunk of the logical tile order so neighbouring tiles hit the same L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
