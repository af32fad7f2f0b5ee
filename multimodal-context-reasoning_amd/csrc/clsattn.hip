// Single-query cross-attention of ClsLayer_lyx (reference modeling/modeling_vcr_chunkalign_v10.py:741-795, called at :857 with one
// [CLS] query per sequence, 8 heads, no mask) over FROZEN bf16 encoder states, REASSOCIATED so that the keys and values are never
// projected:
//     score[h][j] = q_h . (Wk_h x_j + bk_h) = (Wk_h^T q_h) . x_j + const_h      (the constant cancels in the softmax over j)
//     out_h       = sum_j p'[h][j] (Wv_h x_j + bv_h) = Wv_h (sum_j p'[h][j] x_j) + bv_h sum_j p'[h][j]
// The caller forms qt[n][h][:] = Wk_h^T q_h (a few-row GEMM) and applies Wv_h / bv_h to what these kernels return:
//     ctx[n][h][:] = sum_j p'[h][j] x_j   (E wide),   ssum[n][h] = sum_j p'[h][j]   (1 in eval mode; p' = dropout(p), v10:780)
// so the two [N*L, E] x [E, E] projections, their bf16 K / V rows and the K = N*L weight-gradient products of the projected form
// (2.9 ms of a 54 ms step at 128 examples) become two passes over x.  x is read where the encoders left it: up to three row blocks
// (global text rows | chunk-align text rows | chunk-hidden text rows, v10:913) are addressed in place, no concatenated copy.
// One workgroup (4 waves) per sequence; a wave owns key rows w, w+4, ...; a lane holds columns {256 g + 4 lane .. +3} of a row
// (8-byte pieces, G = E / 256 of them); the 8 per-head partial dot products of a row are reduced across the wave by a halving
// butterfly (v_permlane32_swap, v_permlane16_swap, then three xor steps) that leaves head (lane >> 3)'s sum in its 8 lanes.
#include "common.h"

namespace {

constexpr int HH = 8;           // heads of cross_attention_lyx (v10:846)

struct XRows {                  // key rows of one launch: row blocks [N, rows_s, E] concatenated along the key axis
    const bf16* p0; const bf16* p1; const bf16* p2;
    int64_t s0, s1, s2;         // sequence strides (elements)
    int r0, r1, r2;             // rows per block (0 = unused)
    int64_t ld;                 // row stride (elements)
};

__device__ __forceinline__ const bf16* x_row(const XRows& xr, int n, int j) {        // j is wave-uniform
    if (j < xr.r0) return xr.p0 + n * xr.s0 + (int64_t)j * xr.ld;
    j -= xr.r0;
    if (j < xr.r1) return xr.p1 + n * xr.s1 + (int64_t)j * xr.ld;
    j -= xr.r1;
    return xr.p2 + n * xr.s2 + (int64_t)j * xr.ld;
}

// FULL: E == 256 G; otherwise the pieces at columns >= E read as zero (E % 4 == 0)
template <int G, bool FULL>
__device__ __forceinline__ void load_row(const bf16* row, int lane, int E, uint2 (&r)[G]) {
#pragma unroll
    for (int g = 0; g < G; ++g)
        r[g] = (FULL || g * 256 + 4 * lane < E) ? *reinterpret_cast<const uint2*>(row + g * 256 + 4 * lane) : make_uint2(0u, 0u);
}

template <int G, bool FULL>
__device__ __forceinline__ void load_vec(const float* v, int lane, int E, float (&x)[4 * G]) {      // one [E] fp32 vector, lane's columns
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const f32x4 t = (FULL || g * 256 + 4 * lane < E) ? *reinterpret_cast<const f32x4*>(v + g * 256 + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
        x[4 * g] = t[0]; x[4 * g + 1] = t[1]; x[4 * g + 2] = t[2]; x[4 * g + 3] = t[3];
    }
}

template <int G, bool FULL>
__device__ __forceinline__ void store_vec(float* v, int lane, int E, const float (&x)[4 * G]) {
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (FULL || g * 256 + 4 * lane < E)
            *reinterpret_cast<f32x4*>(v + g * 256 + 4 * lane) = f32x4{x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
}

template <int G>
__device__ __forceinline__ void row_f32(const uint2 (&r)[G], float (&x)[4 * G]) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        x[4 * g + 0] = __uint_as_float(r[g].x << 16);
        x[4 * g + 1] = __uint_as_float(r[g].x & 0xffff0000u);
        x[4 * g + 2] = __uint_as_float(r[g].y << 16);
        x[4 * g + 3] = __uint_as_float(r[g].y & 0xffff0000u);
    }
}

// sum over the 64 lanes of v[h] for each of the 8 heads; returns head (lane >> 3)'s total (the same value in its 8 lanes)
__device__ __forceinline__ float reduce8(const float (&v)[HH], int lane) {
    float a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // lanes 0-31 keep heads i, lanes 32-63 heads 4 + i
        const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 4]), false, false);
        a[i] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {       // lane row r (16 lanes) keeps heads 2 r + i
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + 2]), false, false);
        b[i] = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    }
    const bool hi = (lane & 8) != 0;
    float r = (hi ? b[1] : b[0]) + __shfl_xor(hi ? b[0] : b[1], 8, 64);
    r += __shfl_xor(r, 4, 64);
    r += __shfl_xor(r, 2, 64);
    r += __shfl_xor(r, 1, 64);
    return r;
}

// the four waves' [HH][256 G] partial sums -> wave 3 holds the total (sRed: HH * 256 G floats)
template <int G>
__device__ __forceinline__ void sum_waves(float (&acc)[HH][4 * G], float* sRed, int wave, int lane) {
    constexpr int E = 256 * G;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int h = 0; h < HH; ++h)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    f32x4* at = reinterpret_cast<f32x4*>(sRed + h * E + g * 256 + 4 * lane);
                    f32x4 t = {acc[h][4 * g], acc[h][4 * g + 1], acc[h][4 * g + 2], acc[h][4 * g + 3]};
                    if (w > 0) {
                        const f32x4 o = *at;
                        t += o;
                        acc[h][4 * g] = t[0]; acc[h][4 * g + 1] = t[1]; acc[h][4 * g + 2] = t[2]; acc[h][4 * g + 3] = t[3];
                    }
                    if (w < 3) *at = t;
                }
        }
        if (w < 3) __syncthreads();
    }
}

// ---- forward -------------------------------------------------------------------------------------------------------------------
template <int G, int U, bool FULL>
__global__ __launch_bounds__(256) void cls_xattn_fwd_kernel(const float* __restrict__ qt, XRows xr, float* __restrict__ ctx,
                                                            float* __restrict__ ssum, float* __restrict__ probs, int L, int E,
                                                            uint64_t seed, uint64_t offset, uint32_t thr, float keep_scale) {
    extern __shared__ float sm[];
    float* sS = sm;                        // [L][HH] scores -> masked probabilities
    float* sRed = sm + ((L * HH + 3) & ~3);
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    {
        float qv[HH][4 * G];
#pragma unroll
        for (int h = 0; h < HH; ++h) load_vec<G, FULL>(qt + ((int64_t)n * HH + h) * E, lane, E, qv[h]);
        for (int j0 = wave; j0 < L; j0 += 4 * U) {
            uint2 r[U][G];
#pragma unroll
            for (int u = 0; u < U; ++u) load_row<G, FULL>(x_row(xr, n, min(j0 + 4 * u, L - 1)), lane, E, r[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float x[4 * G], part[HH];
                row_f32<G>(r[u], x);
#pragma unroll
                for (int h = 0; h < HH; ++h) {
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < 4 * G; ++c) s = fmaf(qv[h][c], x[c], s);
                    part[h] = s;
                }
                const float tot = reduce8(part, lane);
                const int j = j0 + 4 * u;
                if (j < L && (lane & 7) == 0) sS[j * HH + (lane >> 3)] = tot;
            }
        }
    }
    __syncthreads();
    // softmax over the keys: wave w takes heads 2 w, 2 w + 1
    for (int hh = 2 * wave; hh < 2 * wave + 2; ++hh) {
        float mx = -INFINITY;
        for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sS[j * HH + hh]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int j = lane; j < L; j += 64) { const float e = expf(sS[j * HH + hh] - mx); sS[j * HH + hh] = e; sum += e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = 1.f / sum;
        float kept = 0.f;
        for (int j = lane; j < L; j += 64) {
            const float pj = sS[j * HH + hh] * inv;
            const int64_t at = ((int64_t)n * HH + hh) * L + j;
            // F.dropout on the attention weights (v10:780): same counters as align_attn_fwd_kernel
            const float pd = thr ? drop_apply(pj, seed, offset + (uint64_t)at, thr, keep_scale) : pj;
            sS[j * HH + hh] = pd;
            kept += pd;
            probs[at] = pj;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o, 64);
        if (lane == 0) ssum[(int64_t)n * HH + hh] = kept;
    }
    __syncthreads();
    float acc[HH][4 * G];
#pragma unroll
    for (int h = 0; h < HH; ++h)
#pragma unroll
        for (int c = 0; c < 4 * G; ++c) acc[h][c] = 0.f;
    for (int j0 = wave; j0 < L; j0 += 4 * U) {
        uint2 r[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) load_row<G, FULL>(x_row(xr, n, min(j0 + 4 * u, L - 1)), lane, E, r[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + 4 * u;
            if (j < L) {
                float x[4 * G];
                row_f32<G>(r[u], x);
                const f32x4 pa = *reinterpret_cast<const f32x4*>(sS + j * HH), pb = *reinterpret_cast<const f32x4*>(sS + j * HH + 4);
                const float pw[HH] = {pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3]};
#pragma unroll
                for (int h = 0; h < HH; ++h)
#pragma unroll
                    for (int c = 0; c < 4 * G; ++c) acc[h][c] = fmaf(pw[h], x[c], acc[h][c]);
            }
        }
    }
    sum_waves<G>(acc, sRed, wave, lane);
    if (wave == 3) {
#pragma unroll
        for (int h = 0; h < HH; ++h) store_vec<G, FULL>(ctx + ((int64_t)n * HH + h) * E, lane, E, acc[h]);
    }
}

// ---- backward (x frozen): d qt -------------------------------------------------------------------------------------------------
// dp'[h][j] = dctx_h . x_j + dssum_h;  delta_h = sum_j p'[h][j] dp'[h][j] = dctx_h . ctx_h + dssum_h ssum_h (no pass over the keys);
// dscore[h][j] = p'[h][j] dp'[h][j] - p[h][j] delta_h;  d qt_h = sum_j dscore[h][j] x_j.   One pass over x.
template <int G, int U, bool FULL>
__global__ __launch_bounds__(256) void cls_xattn_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ dssum,
                                                            const float* __restrict__ ctx, const float* __restrict__ ssum,
                                                            const float* __restrict__ probs, XRows xr, float* __restrict__ dqt,
                                                            int L, int E, uint64_t seed, uint64_t offset, uint32_t thr, float keep_scale) {
    extern __shared__ float sm[];
    float* sP = sm;                                  // [L][HH] probabilities
    float* sPd = sm + ((L * HH + 3) & ~3);           // [L][HH] masked probabilities
    float* sRed = sPd + ((L * HH + 3) & ~3);
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int idx = tid; idx < L * HH; idx += 256) {
        const int hh = idx / L, j = idx - hh * L;
        const int64_t at = ((int64_t)n * HH + hh) * L + j;
        const float pj = probs[at];
        sP[j * HH + hh] = pj;
        sPd[j * HH + hh] = thr ? drop_apply(pj, seed, offset + (uint64_t)at, thr, keep_scale) : pj;
    }
    float dv[HH][4 * G];
    float delta, dsh;
    {
        float part[HH];
#pragma unroll
        for (int h = 0; h < HH; ++h) {
            float cv[4 * G], s = 0.f;
            load_vec<G, FULL>(dctx + ((int64_t)n * HH + h) * E, lane, E, dv[h]);
            load_vec<G, FULL>(ctx + ((int64_t)n * HH + h) * E, lane, E, cv);
#pragma unroll
            for (int c = 0; c < 4 * G; ++c) s = fmaf(dv[h][c], cv[c], s);
            part[h] = s;
        }
        const int hl = lane >> 3;
        dsh = dssum[(int64_t)n * HH + hl];
        delta = reduce8(part, lane) + dsh * ssum[(int64_t)n * HH + hl];
    }
    __syncthreads();
    float acc[HH][4 * G];
#pragma unroll
    for (int h = 0; h < HH; ++h)
#pragma unroll
        for (int c = 0; c < 4 * G; ++c) acc[h][c] = 0.f;
    for (int j0 = wave; j0 < L; j0 += 4 * U) {
        uint2 r[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) load_row<G, FULL>(x_row(xr, n, min(j0 + 4 * u, L - 1)), lane, E, r[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + 4 * u;
            if (j < L) {
                float x[4 * G], part[HH];
                row_f32<G>(r[u], x);
#pragma unroll
                for (int h = 0; h < HH; ++h) {
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < 4 * G; ++c) s = fmaf(dv[h][c], x[c], s);
                    part[h] = s;
                }
                const float dp = reduce8(part, lane) + dsh;
                const float dsc = sPd[j * HH + (lane >> 3)] * dp - sP[j * HH + (lane >> 3)] * delta;
#pragma unroll
                for (int h = 0; h < HH; ++h) {
                    const float w = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dsc), 8 * h));
#pragma unroll
                    for (int c = 0; c < 4 * G; ++c) acc[h][c] = fmaf(w, x[c], acc[h][c]);
                }
            }
        }
    }
    sum_waves<G>(acc, sRed, wave, lane);
    if (wave == 3) {
#pragma unroll
        for (int h = 0; h < HH; ++h) store_vec<G, FULL>(dqt + ((int64_t)n * HH + h) * E, lane, E, acc[h]);
    }
}

int fill_rows(XRows& xr, const void* const* x_blocks, const int64_t* seq_strides, const int32_t* rows, int nblocks, int64_t ldx,
              int E, int* L) {
    const bf16* p[3] = {nullptr, nullptr, nullptr};
    int64_t s[3] = {0, 0, 0};
    int r[3] = {0, 0, 0};
    int tot = 0;
    for (int i = 0; i < nblocks; ++i) {
        MODCR_REQUIRE(x_blocks[i] && rows[i] > 0 && seq_strides[i] >= 0, "cls_xattn: bad row block %d", i);
        MODCR_REQUIRE((((uintptr_t)x_blocks[i]) & 7) == 0 && (seq_strides[i] % 4) == 0, "cls_xattn: row block %d is not 8-byte aligned", i);
        p[i] = (const bf16*)x_blocks[i]; s[i] = seq_strides[i]; r[i] = rows[i]; tot += rows[i];
    }
    MODCR_REQUIRE(ldx >= E && (ldx % 4) == 0, "cls_xattn: ldx=%lld", (long long)ldx);
    xr.p0 = p[0]; xr.p1 = p[1] ? p[1] : p[0]; xr.p2 = p[2] ? p[2] : p[0];
    xr.s0 = s[0]; xr.s1 = s[1]; xr.s2 = s[2];
    xr.r0 = r[0]; xr.r1 = r[1]; xr.r2 = r[2];
    xr.ld = ldx;
    *L = tot;
    return MODCR_OK;
}

template <int G, int U, bool FULL>
void allow_lds() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cls_xattn_fwd_kernel<G, U, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cls_xattn_bwd_kernel<G, U, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void configure_once() {                 // write-once: the kernels may take more than the default 64 KB of LDS (long key axes)
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (configured) return;
    allow_lds<3, 4, true>(); allow_lds<4, 2, true>(); allow_lds<1, 4, false>(); allow_lds<2, 4, false>(); allow_lds<3, 4, false>();
    allow_lds<4, 2, false>();
    configured = true;
}

}  // namespace

extern "C" int modcr_cls_xattn_fwd(const float* qt, const void* const* x_blocks, const int64_t* seq_strides, const int32_t* rows,
                                   int32_t nblocks, int64_t ldx, float* ctx, float* ssum, float* probs, int32_t N, int32_t E,
                                   int32_t heads, float p, uint64_t seed, uint64_t offset, modcr_stream_t stream) {
    MODCR_REQUIRE(qt && x_blocks && seq_strides && rows && ctx && ssum && probs, "cls_xattn_fwd: null pointer");
    MODCR_REQUIRE(nblocks >= 1 && nblocks <= 3 && N > 0, "cls_xattn_fwd: bad shape");
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "cls_xattn_fwd: p=%g out of [0, 1)", p);
    if (heads != HH || E < 4 || E > 1024 || (E % 4) != 0) {
        modcr_set_error("cls_xattn_fwd: built for 8 heads and E <= 1024, E %% 4 == 0 (E=%d heads=%d): use the projected form (modcr_align_attn_fwd)", E, heads);
        return MODCR_ERR_UNSUPPORTED;
    }
    XRows xr;
    int L;
    const int rc = fill_rows(xr, x_blocks, seq_strides, rows, nblocks, ldx, E, &L);
    if (rc != MODCR_OK) return rc;
    const uint32_t thr = p > 0.f ? (uint32_t)((double)p * 32768.0 + 0.5) : 0u;
    const float keep_scale = 1.0f / (1.0f - p);
    const int G = (E + 255) / 256;
    const size_t shm = ((size_t)((L * HH + 3) & ~3) + (size_t)HH * 256 * G) * sizeof(float);
    MODCR_REQUIRE(shm <= 160 * 1024, "cls_xattn_fwd: L=%d too long", L);
    configure_once();
#define MODCR_CLSX_FWD(G_, U_, FULL_) \
    hipLaunchKernelGGL((cls_xattn_fwd_kernel<G_, U_, FULL_>), dim3(N), dim3(256), shm, (hipStream_t)stream, qt, xr, ctx, ssum, probs, L, E, seed, offset, thr, keep_scale)
    if (E == 768) MODCR_CLSX_FWD(3, 4, true);
    else if (E == 1024) MODCR_CLSX_FWD(4, 2, true);
    else if (G == 1) MODCR_CLSX_FWD(1, 4, false);
    else if (G == 2) MODCR_CLSX_FWD(2, 4, false);
    else if (G == 3) MODCR_CLSX_FWD(3, 4, false);
    else MODCR_CLSX_FWD(4, 2, false);
#undef MODCR_CLSX_FWD
    return modcr_check_launch("cls_xattn_fwd");
}

extern "C" int modcr_cls_xattn_bwd(const float* dctx, const float* dssum, const float* ctx, const float* ssum, const float* probs,
                                   const void* const* x_blocks, const int64_t* seq_strides, const int32_t* rows, int32_t nblocks,
                                   int64_t ldx, float* dqt, int32_t N, int32_t E, int32_t heads, float p, uint64_t seed,
                                   uint64_t offset, modcr_stream_t stream) {
    MODCR_REQUIRE(dctx && dssum && ctx && ssum && probs && x_blocks && seq_strides && rows && dqt, "cls_xattn_bwd: null pointer");
    MODCR_REQUIRE(nblocks >= 1 && nblocks <= 3 && N > 0, "cls_xattn_bwd: bad shape");
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "cls_xattn_bwd: p=%g out of [0, 1)", p);
    if (heads != HH || E < 4 || E > 1024 || (E % 4) != 0) {
        modcr_set_error("cls_xattn_bwd: built for 8 heads and E <= 1024, E %% 4 == 0 (E=%d heads=%d)", E, heads);
        return MODCR_ERR_UNSUPPORTED;
    }
    XRows xr;
    int L;
    const int rc = fill_rows(xr, x_blocks, seq_strides, rows, nblocks, ldx, E, &L);
    if (rc != MODCR_OK) return rc;
    const uint32_t thr = p > 0.f ? (uint32_t)((double)p * 32768.0 + 0.5) : 0u;
    const float keep_scale = 1.0f / (1.0f - p);
    const int G = (E + 255) / 256;
    const size_t shm = (2 * (size_t)((L * HH + 3) & ~3) + (size_t)HH * 256 * G) * sizeof(float);
    MODCR_REQUIRE(shm <= 160 * 1024, "cls_xattn_bwd: L=%d too long", L);
    configure_once();
#define MODCR_CLSX_BWD(G_, U_, FULL_) \
    hipLaunchKernelGGL((cls_xattn_bwd_kernel<G_, U_, FULL_>), dim3(N), dim3(256), shm, (hipStream_t)stream, dctx, dssum, ctx, ssum, probs, xr, dqt, L, E, seed, offset, thr, keep_scale)
    if (E == 768) MODCR_CLSX_BWD(3, 4, true);
    else if (E == 1024) MODCR_CLSX_BWD(4, 2, true);
    else if (G == 1) MODCR_CLSX_BWD(1, 4, false);
    else if (G == 2) MODCR_CLSX_BWD(2, 4, false);
    else if (G == 3) MODCR_CLSX_BWD(3, 4, false);
    else MODCR_CLSX_BWD(4, 2, false);
#undef MODCR_CLSX_BWD
    return modcr_check_launch("cls_xattn_bwd");
}
