// HBM-bound row kernels of the ModCR hot path: LayerNorm (+residual), embedding gather + LN,
// casts, mask bit-packing, the single-query alignment attention and the multiple-choice CE.
// One wave (64 lanes) owns one row; 16-byte vector accesses; wave shuffles for the reductions.
#include <stdarg.h>

#include "common.h"

// ---- error plumbing (shared by all translation units) ---------------------------------------
static thread_local char g_err[512] = "";
void modcr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int modcr_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        modcr_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return MODCR_ERR_LAUNCH;
    }
    return MODCR_OK;
}
extern "C" int modcr_version(void) { return MODCR_VERSION; }
extern "C" const char* modcr_last_error(void) { return g_err; }

namespace {

constexpr int MAXV = 4;  // H <= 64*4*MAXV = 1024

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Vec4<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[4]) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[4]) {
        bf16x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = (bf16)v[i];
        *reinterpret_cast<bf16x4*>(p) = t;
    }
};

template <> struct Vec4<_Float16> {        // IEEE-half rows: the GEMM output a LayerNorm pass reads back (MODCR_F16)
    static __device__ __forceinline__ void load(const _Float16* p, float (&v)[4]) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 t = *reinterpret_cast<const h4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
    }
    static __device__ __forceinline__ void store(_Float16* p, const float (&v)[4]) {     // saturating: +-65504, never inf
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = (_Float16)((v[i] == v[i]) ? __builtin_amdgcn_fmed3f(v[i], -65504.0f, 65504.0f) : v[i]);   // NaN stays NaN (gemm.hip cvt16)
        *reinterpret_cast<h4*>(p) = t;
    }
};

// normalise the row held in v[][] (H = 4*64*nv elements spread over the wave) and store it
// (thr != 0: nn.Dropout on the normalised row before the store, counters ctr0 + column -- BertEmbeddings, a_bert:209-210)
template <typename TO>
__device__ __forceinline__ void ln_finish(float (&v)[MAXV][4], int nv, int H, int lane,
                                          const float* gamma, const float* beta, float eps, TO* out,
                                          uint64_t seed = 0, uint64_t ctr0 = 0, uint32_t thr = 0, float scale = 1.f) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && (lane + 64 * i) * 4 < H) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
        if (i < nv && (lane + 64 * i) * 4 < H)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < H) {
            float g[4], b[4], o[4];
            Vec4<float>::load(gamma + c, g);
            Vec4<float>::load(beta + c, b);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
            if (thr) drop_apply4(o, seed, ctr0 + (uint64_t)c, thr, scale);
            Vec4<TO>::store(out + c, o);
        }
    }
}

template <typename TI, typename TR, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* x, const TR* res, const float* gamma,
                                                        const float* beta, float eps, TO* y, int64_t M,
                                                        int H, int rpg, int64_t gstride) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nv = (H + 255) / 256;
    float v[MAXV][4];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < H) {
            Vec4<TI>::load(x + m * H + c, v[i]);
            if (res) {
                float r[4];
                Vec4<TR>::load(res + m * H + c, r);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] += r[j];
            }
        }
    }
    const int64_t orow = rpg > 0 ? (m / rpg) * gstride + (m % rpg) : m;
    ln_finish<TO>(v, nv, H, lane, gamma, beta, eps, y + orow * H);
}

template <typename TO>
__global__ __launch_bounds__(256) void embed_ln_kernel(const int64_t* ids, const int64_t* tts,
                                                       const int64_t* pos_ids, const float* word,
                                                       const float* pos, const float* type,
                                                       const float* gamma, const float* beta, float eps,
                                                       TO* out, int N, int T, int H, int64_t seq_stride,
                                                       int vocab, int max_pos, int type_vocab,
                                                       uint64_t seed, uint64_t offset, uint32_t thr, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)N * T) return;
    const int n = (int)(row / T), t = (int)(row % T);
    int64_t id = ids[row];
    int64_t tt = tts ? tts[row] : 0;
    int64_t ps = pos_ids ? pos_ids[row] : t;
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);          // clamp: never read outside the tables
    tt = tt < 0 ? 0 : (tt >= type_vocab ? type_vocab - 1 : tt);
    ps = ps < 0 ? 0 : (ps >= max_pos ? max_pos - 1 : ps);
    const int nv = (H + 255) / 256;
    float v[MAXV][4];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < H) {
            float a[4], b[4], d[4];
            Vec4<float>::load(word + id * H + c, a);
            Vec4<float>::load(type + tt * H + c, b);
            Vec4<float>::load(pos + ps * H + c, d);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = (a[j] + b[j]) + d[j];   // a_bert:203-206 order
        }
    }
    const int64_t at = ((int64_t)n * seq_stride + t) * H;
    ln_finish<TO>(v, nv, H, lane, gamma, beta, eps, out + at, seed, offset + (uint64_t)at, thr, scale);
}

template <typename TO>
__global__ void cast_pad_kernel(const float* src, int64_t lds_, TO* dst, int64_t ldd, int64_t M, int K,
                                int Kp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * Kp) return;
    const int64_t m = i / Kp;
    const int k = (int)(i % Kp);
    dst[m * ldd + k] = from_f32<TO>(k < K ? src[m * lds_ + k] : 0.f);
}

// the same, eight output columns (one 16-byte bf16 store) per thread: the step's first kernel casts the [N R, 2054] fp32 region
// features into the 64-padded bf16 operand of the region-embedding GEMM (51200 rows: 421 MB read, 223 MB written) -- the scalar form
// above, a 64-bit division and a 2-byte store per element, ran it in 266 us of the 49.5 ms step; this one in 147 us = 4.4 TB/s.  (Round 5:
// a wave-per-row form with lane-contiguous 8-byte pieces -- 512 contiguous bytes per load instruction instead of 32-byte strides between
// lanes -- measured the same 145 us: the pass is at the rate of its streams.)  Needs Kp % 8 == 0, even
// source row stride (8-byte loads; the 2054-float rows are not 16-byte aligned) and a 16-byte aligned destination.
__global__ __launch_bounds__(256) void cast_pad8_kernel(const float* __restrict__ src, int64_t lds_, bf16* __restrict__ dst, int64_t ldd, int64_t M,
                                                        int K, int Kp) {
    const int cpr = Kp >> 3;                                         // 8-column chunks per row
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * cpr) return;
    const int64_t m = i / cpr;
    const int k0 = (int)(i - m * cpr) * 8;
    const float* sp = src + m * lds_ + k0;
    float v[8];
    if (k0 + 8 <= K) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float2 t = *reinterpret_cast<const float2*>(sp + 2 * e);
            v[2 * e] = t.x; v[2 * e + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = k0 + e < K ? sp[e] : 0.f;
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
    *reinterpret_cast<bf16x8*>(dst + m * ldd + k0) = o;
}

// fp32 -> three bf16 terms per element so that a bf16 MFMA GEMM over the tripled K reproduces the
// fp32 product to ~2^-16: x = hi + lo (hi = bf16(x), lo = bf16(x - hi)).
//   mode 0 (activations): [hi | lo | hi]     mode 1 (weights): [hi | hi | lo]
// => sum over 3K of a3*w3 = a_hi*w_hi + a_lo*w_hi + a_hi*w_lo.
__global__ void split3_kernel(const float* src, int64_t lds_, bf16* dst, int64_t ldd, int64_t M, int K, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * K) return;
    const int64_t m = i / K;
    const int k = (int)(i % K);
    const float x = src[m * lds_ + k];
    const bf16 hi = (bf16)x;
    const bf16 lo = (bf16)(x - (float)hi);
    bf16* row = dst + m * ldd;
    row[k] = hi;
    row[K + k] = mode == 0 ? lo : hi;
    row[2 * K + k] = mode == 0 ? hi : lo;
}

template <typename TI, typename TO>
__global__ void convert_kernel(const TI* src, TO* dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = from_f32<TO>(to_f32(src[i]));
}

// one wave packs 64 mask words... simpler: one thread per output word
__global__ void pack_bits_kernel(const float* mask, uint32_t* bits, int64_t rows, int L, int LW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * LW) return;
    const int64_t row = i / LW;
    const int w = (int)(i % LW);
    uint32_t b = 0;
    for (int j = 0; j < 32; ++j) {
        const int c = w * 32 + j;
        if (c < L && mask[row * L + c] != 0.f) b |= 1u << j;
    }
    bits[i] = b;
}

// k short sequences as ONE row block of k S rows: key j of the block is visible to query i iff both lie in the same sequence and
// the key is not padding.  One thread per (block, row, word).
__global__ void packed_mask_kernel(const float* key_mask, uint32_t* bits, int64_t NB, int S, int k) {
    const int L = S * k, LW = (L + 31) / 32;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NB * L * LW) return;
    const int w = (int)(idx % LW);
    const int i = (int)((idx / LW) % L);
    const int64_t v = idx / ((int64_t)LW * L);
    const int si = i / S;
    uint32_t b = 0;
    for (int j = 0; j < 32; ++j) {
        const int c = w * 32 + j;
        if (c < L && c / S == si && key_mask[(v * k + si) * S + (c - si * S)] != 0.f) b |= 1u << j;
    }
    bits[idx] = b;
}

// v10:179-206 as a predicate.  One thread per (n, query row i, word w).
__global__ void phase_mask_kernel(const float* input_mask, const float* chunk_mask, uint32_t* bits,
                                  int N, int T, int R, int phase) {
    const int S = T + R, LW = (S + 31) / 32;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * S * LW) return;
    const int w = (int)(idx % LW);
    const int i = (int)((idx / LW) % S);
    const int n = (int)(idx / ((int64_t)LW * S));
    uint32_t b = 0;
    for (int jj = 0; jj < 32; ++jj) {
        const int j = w * 32 + jj;
        if (j >= S) break;
        bool see;
        if (i < T) {
            see = (j < T) ? chunk_mask[((int64_t)n * T + i) * T + j] != 0.f
                          : input_mask[(int64_t)n * S + j] != 0.f;
        } else if (phase == 1) {
            see = (j < T) ? false : input_mask[(int64_t)n * S + j] != 0.f;
        } else {
            see = (j == i);
        }
        if (see) b |= 1u << jj;
    }
    bits[idx] = b;
}

// ---- chunk-mean query (v10:66-78), standalone form: one block per (n, 64-column slab) ----------
// (the trainable path's forward and the adjoint in its backward; the fused attention kernel has its own, attn4_chunk_mean).
// The [T][64] slab and the sequence's ids are staged in LDS with all loads in flight, the means are formed from the LDS copy
// (any id pattern: token t averages every token with its id) and written straight back.  The first version walked the rows of
// one column per thread with dependent global loads inside an O(T^2) scan: 700 us per call at N = 512, T = 80.
template <typename T>
__global__ __launch_bounds__(256) void chunk_mean_q_kernel(T* q, int64_t row_stride, int64_t seq_stride,
                                                          const int32_t* chunk_id, int Tn, int H) {
    extern __shared__ float s_cm[];
    const int TP = (Tn + 3) & ~3;
    int* s_cid = reinterpret_cast<int*>(s_cm);
    int* s_lo = s_cid + TP;                                 // first / last member of token t's chunk when the chunk is ONE
    int* s_hi = s_lo + TP;                                  // contiguous run (the data format); lo = -1: scan all tokens
    float* tile = s_cm + 3 * TP;                            // [Tn][64]
    const int n = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    T* base = q + (int64_t)n * seq_stride + c0;
    for (int t = tid; t < Tn; t += 256) s_cid[t] = chunk_id[(int64_t)n * Tn + t];
    for (int idx = tid; idx < Tn * 64; idx += 256) {
        const int t = idx >> 6, cc = idx & 63;
        tile[idx] = (c0 + cc < H) ? to_f32(base[(int64_t)t * row_stride + cc]) : 0.f;
    }
    __syncthreads();
    for (int t = tid; t < Tn; t += 256) {
        const int id = s_cid[t];
        int lo = Tn, hi = -1, cnt = 0;
        if (id >= 0)
            for (int u = 0; u < Tn; ++u)
                if (s_cid[u] == id) { lo = min(lo, u); hi = max(hi, u); ++cnt; }
        const bool run = cnt > 0 && hi - lo + 1 == cnt;
        s_lo[t] = run ? lo : -1;
        s_hi[t] = run ? hi : -1;
    }
    __syncthreads();
    const int cc = tid & 63;
    if (c0 + cc >= H) return;
    for (int t = tid >> 6; t < Tn; t += 4) {
        const int id = s_cid[t];
        if (id < 0) continue;
        float s = 0.f;
        int cnt = 0;
        if (s_lo[t] >= 0) {
            for (int u = s_lo[t]; u <= s_hi[t]; ++u) s += tile[u * 64 + cc];
            cnt = s_hi[t] - s_lo[t] + 1;
        } else {
            for (int u = 0; u < Tn; ++u)
                if (s_cid[u] == id) { s += tile[u * 64 + cc]; ++cnt; }
        }
        if (cnt > 1) base[(int64_t)t * row_stride + cc] = from_f32<T>(s / (float)cnt);
    }
}

// ---- alignment attention (v10:741-795 with tgt_len = 1): one 256-thread workgroup per sequence ---------
// q / out / dq are fp32 [N,E] (the CLS path of the trainable head); k, v, dk, dv are T [N,L,E]
// (projected encoder states).  scores = (q*scale).k  (v10:710: the scaling sits on the query).
// HBM-bound: K and V rows are read (and dK, dV rows written) once, as whole rows in 16-byte (bf16) /
// 32-byte (fp32) pieces: thread (slot, c) owns feature chunk c (8 features, C = E/8 chunks per row) of the
// keys slot, slot + KS, ...; per-(key, chunk) partial dot products go through LDS and are summed per head.
template <typename T> struct Vec8;
template <> struct Vec8<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) {
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) {
        bf16x8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
        *reinterpret_cast<bf16x8*>(p) = t;
    }
};
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};

// sPart[L][C] partial dots -> sS[L][heads] per-head sums
__device__ __forceinline__ void align_reduce_heads(const float* sPart, float* sS, int L, int C, int heads) {
    const int cph = C / heads;                                  // chunks per head
    for (int idx = threadIdx.x; idx < L * heads; idx += 256) {
        const int j = idx / heads, h = idx % heads;
        float s = 0.f;
        for (int c = 0; c < cph; ++c) s += sPart[j * C + h * cph + c];
        sS[idx] = s;
    }
}

// counter-based dropout decisions: drop_words / drop_field / drop_apply / drop_apply4 live in common.h (shared with the
// LayerNorm epilogue of gemm.hip)

template <typename T>
__global__ __launch_bounds__(256) void align_attn_fwd_kernel(const float* q, const T* k, const T* v,
                                                            int64_t ldkv, float* out, float* probs, int L,
                                                            int E, int heads, float scale, uint64_t seed, uint64_t offset,
                                                            uint32_t thr, float keep_scale, const float* key_bias, int LB) {
    extern __shared__ float sm[];
    const int C = E / 8, KS = 256 / C, cph = C / heads;
    float* sPart = sm;                     // [LB][C] (one block of keys); later [KS][E] output partials
    float* sS = sm + max(LB * C, KS * E);  // [L][heads] scores -> probabilities
    const int n = blockIdx.x, tid = threadIdx.x;
    const int slot = tid / C, c = tid % C;
    const bool active = slot < KS;
    const int h = c / cph;
    const T* kb = k + (int64_t)n * L * ldkv + c * 8;
    const T* vb = v + (int64_t)n * L * ldkv + c * 8;
    float qv[8];
    if (active) {
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[i] = q[(int64_t)n * E + c * 8 + i] * scale;
    }
    // scores in blocks of LB keys (LB % (4 KS) == 0): the [LB][C] partial products of a block, then their per-head sums
    for (int jb = 0; jb < L; jb += LB) {
        const int nb = min(LB, L - jb);
        if (active) {
            for (int j0 = jb + slot; j0 < jb + nb; j0 += 4 * KS) {        // four rows in flight per thread
                float kv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) Vec8<T>::load(kb + (int64_t)min(j0 + u * KS, L - 1) * ldkv, kv[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u * KS;
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s = fmaf(qv[i], kv[u][i], s);
                    if (j < jb + nb) sPart[(j - jb) * C + c] = s;
                }
            }
        }
        __syncthreads();
        align_reduce_heads(sPart, sS + jb * heads, nb, C, heads);
        __syncthreads();
    }
    // softmax over the keys, 32 lanes per head (heads <= 8)
    {
        const int hh = tid >> 5, l32 = tid & 31;
        if (hh < heads) {
            float mx = -INFINITY;
            if (key_bias)       // additive key mask (ClsLayer2's word_mask, v10:823): same for every head
                for (int j = l32; j < L; j += 32) sS[j * heads + hh] += key_bias[(int64_t)n * L + j];
            for (int j = l32; j < L; j += 32) mx = fmaxf(mx, sS[j * heads + hh]);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            float sum = 0.f;
            for (int j = l32; j < L; j += 32) { const float e = expf(sS[j * heads + hh] - mx); sS[j * heads + hh] = e; sum += e; }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            const float inv = 1.f / sum;
            for (int j = l32; j < L; j += 32) {
                const float pj = sS[j * heads + hh] * inv;
                const int64_t at = ((int64_t)n * heads + hh) * L + j;
                // F.dropout on the attention weights (v10:780): the value product uses the masked weights, the
                // backward's softmax the unmasked ones
                sS[j * heads + hh] = thr ? drop_apply(pj, seed, offset + (uint64_t)at, thr, keep_scale) : pj;
                if (probs) probs[at] = pj;
            }
        }
    }
    __syncthreads();
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        for (int j0 = slot; j0 < L; j0 += 4 * KS) {
            float vv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) Vec8<T>::load(vb + (int64_t)min(j0 + u * KS, L - 1) * ldkv, vv[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * KS;
                const float pj = j < L ? sS[j * heads + h] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaf(pj, vv[u][i], acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sPart[slot * E + c * 8 + i] = acc[i];
    }
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        float o = 0.f;
        for (int s2 = 0; s2 < KS; ++s2) o += sPart[s2 * E + e];
        out[(int64_t)n * E + e] = o;
    }
}

// backward: dv[j] = p_j dout; dp_j = dout.v_j; ds_j = p_j (dp_j - sum_i p_i dp_i);
// dq = scale * sum_j ds_j k_j; dk_j = ds_j * scale * q.
template <typename T>
__global__ __launch_bounds__(256) void align_attn_bwd_kernel(const float* dout, const float* q, const T* k,
                                                            const T* v, int64_t ldkv, const float* probs,
                                                            float* dq, T* dk, T* dv, int64_t lddkv, int L, int E,
                                                            int heads, float scale, uint64_t seed, uint64_t offset,
                                                            uint32_t thr, float keep_scale, int LB) {
    extern __shared__ float sm[];
    const int C = E / 8, KS = 256 / C, cph = C / heads;
    float* sPart = sm;                     // [LB][C] (one block of keys); later [KS][E] dq partials
    float* sS = sm + max(LB * C, KS * E);  // [L][heads] dp -> ds
    float* sP = sS + (size_t)L * heads;    // [L][heads] probabilities
    const int n = blockIdx.x, tid = threadIdx.x;
    const int slot = tid / C, c = tid % C;
    const bool active = slot < KS;
    const int h = c / cph;
    const T* kb = k + (int64_t)n * L * ldkv + c * 8;
    const T* vb = v + (int64_t)n * L * ldkv + c * 8;
    T* dkb = dk + (int64_t)n * L * lddkv + c * 8;
    T* dvb = dv + (int64_t)n * L * lddkv + c * 8;
    for (int idx = tid; idx < L * heads; idx += 256) {
        const int j = idx / heads, hh = idx % heads;
        sP[idx] = probs[((int64_t)n * heads + hh) * L + j];
    }
    float dov[8], qv[8];
    if (active) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dov[i] = dout[(int64_t)n * E + c * 8 + i];
            qv[i] = q[(int64_t)n * E + c * 8 + i] * scale;
        }
    }
    for (int jb = 0; jb < L; jb += LB) {                    // dp = dout . v_j in blocks of LB keys (as the forward's scores)
        const int nb = min(LB, L - jb);
        if (active) {
            for (int j0 = jb + slot; j0 < jb + nb; j0 += 4 * KS) {
                float vv[4][8];
#pragma unroll
                for (int u = 0; u < 4; ++u) Vec8<T>::load(vb + (int64_t)min(j0 + u * KS, L - 1) * ldkv, vv[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u * KS;
                    float s = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s = fmaf(dov[i], vv[u][i], s);
                    if (j < jb + nb) sPart[(j - jb) * C + c] = s;
                }
            }
        }
        __syncthreads();
        align_reduce_heads(sPart, sS + jb * heads, nb, C, heads);
        __syncthreads();
    }
    {
        const int hh = tid >> 5, l32 = tid & 31;
        if (hh < heads) {
            float dot = 0.f;
            if (thr)        // gradient of the unmasked weights = mask * (gradient of the masked ones)
                for (int j = l32; j < L; j += 32)
                    sS[j * heads + hh] = drop_apply(sS[j * heads + hh], seed, offset + (uint64_t)(((int64_t)n * heads + hh) * L + j), thr, keep_scale);
            for (int j = l32; j < L; j += 32) dot += sP[j * heads + hh] * sS[j * heads + hh];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
            for (int j = l32; j < L; j += 32) sS[j * heads + hh] = sP[j * heads + hh] * (sS[j * heads + hh] - dot);
        }
    }
    __syncthreads();
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
        for (int j0 = slot; j0 < L; j0 += 4 * KS) {
            float kv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) Vec8<T>::load(kb + (int64_t)min(j0 + u * KS, L - 1) * ldkv, kv[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * KS;
                if (j < L) {
                    float o1[8], o2[8];
                    float pj = sP[j * heads + h];
                    const float ds = sS[j * heads + h];
                    if (thr) pj = drop_apply(pj, seed, offset + (uint64_t)(((int64_t)n * heads + h) * L + j), thr, keep_scale);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[i] = fmaf(ds, kv[u][i], acc[i]);
                        o1[i] = pj * dov[i];
                        o2[i] = ds * qv[i];
                    }
                    Vec8<T>::store(dvb + (int64_t)j * lddkv, o1);
                    Vec8<T>::store(dkb + (int64_t)j * lddkv, o2);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) sPart[slot * E + c * 8 + i] = acc[i];
    }
    __syncthreads();
    for (int e = tid; e < E; e += 256) {
        float g = 0.f;
        for (int s2 = 0; s2 < KS; ++s2) g += sPart[s2 * E + e];
        dq[(int64_t)n * E + e] = g * scale;
    }
}

// ---- multiple-choice CE fwd+bwd: one thread per example, one block (B <= a few thousand) --------
__global__ __launch_bounds__(256) void mc_ce_kernel(const float* logits, const float* label, float* loss,
                                                    float* dlogits, const float* grad_scale, int B, int C) {
    __shared__ float part[4];
    float local = 0.f;
    const float gs = (grad_scale ? *grad_scale : 1.0f) / (float)B;
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* z = logits + (int64_t)b * C;
        const float* y = label + (int64_t)b * C;
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, z[c]);
        float se = 0.f, ysum = 0.f;
        for (int c = 0; c < C; ++c) { se += expf(z[c] - mx); ysum += y[c]; }
        const float lse = mx + logf(se);
        for (int c = 0; c < C; ++c) {
            local -= y[c] * (z[c] - lse);
            if (dlogits) dlogits[(int64_t)b * C + c] = (expf(z[c] - lse) * ysum - y[c]) * gs;
        }
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0 && loss) *loss = (part[0] + part[1] + part[2] + part[3]) / (float)B;
}

// ---- LayerNorm backward, H <= 1024: one wave per row, RPW rows per wave; the row lives in registers (one read of
// x / residual / dy), dgamma / dbeta partials stay in registers over the wave's rows, are summed over the block's
// four waves in LDS and leave as ONE atomic per column per block (the per-element atomics of the kernel below cost
// 917 us at M = 27136, H = 1024: 55 M atomics on 2048 addresses).
template <int NC>      // columns per lane: H <= 64 NC
__global__ __launch_bounds__(256) void layernorm_bwd_rows_kernel(const float* dY, const float* pre, const float* res,
                                                                 const float* gamma, float eps, float* dX,
                                                                 float* dgamma, float* dbeta, int64_t M, int H, int rpw) {
    __shared__ float sPart[2][4][64 * NC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pg[NC], pb[NC], gm[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { pg[k] = 0.f; pb[k] = 0.f; const int c = lane + 64 * k; gm[k] = c < H ? gamma[c] : 0.f; }
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * rpw;
    for (int64_t m = m0; m < m0 + rpw && m < M; ++m) {
        float xr[NC], dy[NC];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            xr[k] = 0.f; dy[k] = 0.f;
            if (c < H) { xr[k] = pre[m * H + c] + (res ? res[m * H + c] : 0.f); dy[k] = dY[m * H + c]; s += xr[k]; }
        }
        const float mean = wave_sum(s) / (float)H;
        float qv = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) if (lane + 64 * k < H) { const float d = xr[k] - mean; qv += d * d; }
        const float rstd = rsqrtf(wave_sum(qv) / (float)H + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            xr[k] = (xr[k] - mean) * rstd;            // x_hat (0 * rstd... for c >= H: unused)
            const float g = dy[k] * gm[k];
            a += g; b += g * xr[k];
        }
        a = wave_sum(a) / (float)H;
        b = wave_sum(b) / (float)H;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int c = lane + 64 * k;
            if (c < H) {
                dX[m * H + c] = rstd * (dy[k] * gm[k] - a - xr[k] * b);
                pg[k] += dy[k] * xr[k];
                pb[k] += dy[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) { sPart[0][wave][lane + 64 * k] = pg[k]; sPart[1][wave][lane + 64 * k] = pb[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        if (dgamma) atomicAdd(dgamma + c, sPart[0][0][c] + sPart[0][1][c] + sPart[0][2][c] + sPart[0][3][c]);
        if (dbeta) atomicAdd(dbeta + c, sPart[1][0][c] + sPart[1][1][c] + sPart[1][2][c] + sPart[1][3][c]);
    }
}

// ---- LayerNorm backward (fp32, any H): one wave per row + atomics for dgamma/dbeta ---------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* dY, const float* pre, const float* res,
                                                            const float* gamma, float eps, float* dX,
                                                            float* dgamma, float* dbeta, int64_t M, int H) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* xp = pre + m * H;
    const float* rp = res ? res + m * H : nullptr;
    const float* dy = dY + m * H;
    auto X = [&](int c) { return rp ? xp[c] + rp[c] : xp[c]; };
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s += X(c);
    const float mean = wave_sum(s) / (float)H;
    float qv = 0.f;
    for (int c = lane; c < H; c += 64) { const float d = X(c) - mean; qv += d * d; }
    const float rstd = rsqrtf(wave_sum(qv) / (float)H + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < H; c += 64) {
        const float xh = (X(c) - mean) * rstd;
        const float g = dy[c] * gamma[c];
        a += g;
        b += g * xh;
    }
    a = wave_sum(a) / (float)H;
    b = wave_sum(b) / (float)H;
    for (int c = lane; c < H; c += 64) {
        const float xh = (X(c) - mean) * rstd;
        dX[m * H + c] = rstd * (dy[c] * gamma[c] - a - xh * b);
        if (dgamma) atomicAdd(dgamma + c, dy[c] * xh);
        if (dbeta) atomicAdd(dbeta + c, dy[c]);
    }
}

__global__ void act_bwd_kernel(const float* dact, const float* pre, float* dpre, int64_t n, int act) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = pre[i];
    float g = 1.f;
    if (act == MODCR_ACT_GELU) {
        g = 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) +
            x * 0.39894228040143267794f * expf(-0.5f * x * x);
    } else if (act == MODCR_ACT_TANH) {
        const float t = tanhf(x);
        g = 1.f - t * t;
    }
    dpre[i] = dact[i] * g;
}

inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

extern "C" int modcr_layernorm_fwd(const void* x, int32_t in_dtype, const void* residual,
                                   int32_t res_dtype, const float* gamma, const float* beta, float eps,
                                   void* y, int32_t out_dtype, int64_t M, int32_t H,
                                   int32_t rows_per_group, int64_t group_stride, modcr_stream_t stream) {
    MODCR_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
    MODCR_REQUIRE(M > 0 && H > 0 && (H % 4) == 0 && H <= 256 * MAXV, "layernorm_fwd: H=%d must be a multiple of 4 and <= 1024", H);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(blocks_for(M, 4)), blk(256);
    const int key = in_dtype * 4 + (residual ? res_dtype : in_dtype) * 2 + out_dtype;
#define LN_CASE(K, TI, TR, TO)                                                                          \
    case K:                                                                                             \
        hipLaunchKernelGGL((layernorm_kernel<TI, TR, TO>), grid, blk, 0, st, (const TI*)x,              \
                           (const TR*)residual, gamma, beta, eps, (TO*)y, M, H, rows_per_group,         \
                           group_stride);                                                               \
        break;
    if (in_dtype == MODCR_F16) {            // fp16 pre-LayerNorm rows of the bf16 path: bf16 (or no) residual, bf16 output
        MODCR_REQUIRE((!residual || res_dtype == MODCR_BF16) && out_dtype == MODCR_BF16, "layernorm_fwd: fp16 rows need a bf16 residual and output");
        hipLaunchKernelGGL((layernorm_kernel<_Float16, bf16, bf16>), grid, blk, 0, st, (const _Float16*)x, (const bf16*)residual, gamma, beta,
                           eps, (bf16*)y, M, H, rows_per_group, group_stride);
        return modcr_check_launch("layernorm");
    }
    switch (key) {
        LN_CASE(0, bf16, bf16, bf16) LN_CASE(1, bf16, bf16, float) LN_CASE(2, bf16, float, bf16)
        LN_CASE(3, bf16, float, float) LN_CASE(4, float, bf16, bf16) LN_CASE(5, float, bf16, float)
        LN_CASE(6, float, float, bf16) LN_CASE(7, float, float, float)
        default: MODCR_REQUIRE(false, "layernorm_fwd: bad dtypes");
    }
#undef LN_CASE
    return modcr_check_launch("layernorm");
}

namespace { inline uint32_t drop_threshold(float p) { return (uint32_t)((double)p * 32768.0 + 0.5); } }     // 15-bit uniforms (common.h: drop_words)

// BertEmbeddings.forward (a_bert:195-211) incl. its dropout (p > 0: training mode): the decision of element (n, t, c) is counter
// offset + (n * seq_stride + t) * H + c, i.e. the flat index inside the caller's [N, seq_stride, H] buffer -- the mask
// modcr_dropout(out, n = N * seq_stride * H, seed, offset) would apply to these rows.
extern "C" int modcr_embed_ln_dropout_fwd(const int64_t* input_ids, const int64_t* token_type_ids,
                                          const int64_t* position_ids, const float* word, const float* pos,
                                          const float* type, const float* gamma, const float* beta, float eps,
                                          void* out, int32_t N, int32_t T, int32_t H, int64_t seq_stride,
                                          int32_t vocab, int32_t max_pos, int32_t type_vocab, int32_t out_dtype,
                                          float p, uint64_t seed, uint64_t offset, modcr_stream_t stream) {
    MODCR_REQUIRE(input_ids && word && pos && type && gamma && beta && out, "embed_ln_fwd: null pointer");
    MODCR_REQUIRE(N > 0 && T > 0 && (H % 4) == 0 && H <= 256 * MAXV, "embed_ln_fwd: bad shape");
    MODCR_REQUIRE(position_ids || T <= max_pos, "embed_ln_fwd: T=%d exceeds max_position_embeddings=%d", T, max_pos);
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "embed_ln_fwd: p=%g out of [0, 1)", p);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(blocks_for((int64_t)N * T, 4)), blk(256);
    const uint32_t thr = p > 0.f ? drop_threshold(p) : 0u;
    const float scale = 1.0f / (1.0f - p);
    if (out_dtype == MODCR_BF16)
        hipLaunchKernelGGL((embed_ln_kernel<bf16>), grid, blk, 0, st, input_ids, token_type_ids, position_ids,
                           word, pos, type, gamma, beta, eps, (bf16*)out, N, T, H, seq_stride, vocab, max_pos, type_vocab, seed, offset, thr, scale);
    else
        hipLaunchKernelGGL((embed_ln_kernel<float>), grid, blk, 0, st, input_ids, token_type_ids, position_ids,
                           word, pos, type, gamma, beta, eps, (float*)out, N, T, H, seq_stride, vocab, max_pos, type_vocab, seed, offset, thr, scale);
    return modcr_check_launch("embed_ln");
}

extern "C" int modcr_embed_ln_fwd(const int64_t* input_ids, const int64_t* token_type_ids,
                                  const int64_t* position_ids, const float* word, const float* pos,
                                  const float* type, const float* gamma, const float* beta, float eps,
                                  void* out, int32_t N, int32_t T, int32_t H, int64_t seq_stride,
                                  int32_t vocab, int32_t max_pos, int32_t type_vocab, int32_t out_dtype,
                                  modcr_stream_t stream) {
    return modcr_embed_ln_dropout_fwd(input_ids, token_type_ids, position_ids, word, pos, type, gamma, beta, eps, out, N, T, H, seq_stride,
                                      vocab, max_pos, type_vocab, out_dtype, 0.f, 0, 0, stream);
}

extern "C" int modcr_cast_pad(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t M,
                              int32_t K, int32_t Kp, int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(src && dst && M > 0 && K > 0 && Kp >= K && ldd >= Kp && lds_ >= K, "cast_pad: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(blocks_for(M * Kp, 256)), blk(256);
    if (dtype == MODCR_BF16 && (Kp % 8) == 0 && (lds_ % 2) == 0 && (ldd % 8) == 0 && modcr_aligned16(dst) && (((uintptr_t)src) & 7) == 0)
        hipLaunchKernelGGL(cast_pad8_kernel, dim3(blocks_for(M * (Kp / 8), 256)), blk, 0, st, src, lds_, (bf16*)dst, ldd, M, K, Kp);
    else if (dtype == MODCR_BF16)
        hipLaunchKernelGGL((cast_pad_kernel<bf16>), grid, blk, 0, st, src, lds_, (bf16*)dst, ldd, M, K, Kp);
    else
        hipLaunchKernelGGL((cast_pad_kernel<float>), grid, blk, 0, st, src, lds_, (float*)dst, ldd, M, K, Kp);
    return modcr_check_launch("cast_pad");
}

extern "C" int modcr_convert(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n,
                             modcr_stream_t stream) {
    MODCR_REQUIRE(src && dst && n > 0, "convert: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(blocks_for(n, 256)), blk(256);
    if (src_dtype == MODCR_F32 && dst_dtype == MODCR_BF16)
        hipLaunchKernelGGL((convert_kernel<float, bf16>), grid, blk, 0, st, (const float*)src, (bf16*)dst, n);
    else if (src_dtype == MODCR_BF16 && dst_dtype == MODCR_F32)
        hipLaunchKernelGGL((convert_kernel<bf16, float>), grid, blk, 0, st, (const bf16*)src, (float*)dst, n);
    else if (src_dtype == MODCR_F32 && dst_dtype == MODCR_F32)
        hipLaunchKernelGGL((convert_kernel<float, float>), grid, blk, 0, st, (const float*)src, (float*)dst, n);
    else
        hipLaunchKernelGGL((convert_kernel<bf16, bf16>), grid, blk, 0, st, (const bf16*)src, (bf16*)dst, n);
    return modcr_check_launch("convert");
}

// ---- several conversions in one launch: a trainable layer's weights go fp32 (the optimizer's copy) -> bf16 (the GEMMs' copy)
// after every optimizer step; per tensor that was a launch of a few microseconds each (and a torch.cat for q | k | v: here the
// three land side by side because the caller hands consecutive destinations)
namespace {
struct ConvSegs {
    const void* src[8];
    void* dst[8];
    int64_t n[8];
};
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void convert_segments_kernel(ConvSegs sg) {
    const int k = blockIdx.y;
    const TI* src = reinterpret_cast<const TI*>(sg.src[k]);
    TO* dst = reinterpret_cast<TO*>(sg.dst[k]);
    const int64_t n = sg.n[k];
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n && ((reinterpret_cast<uintptr_t>(src + i) & (4 * sizeof(TI) - 1)) == 0) &&
            ((reinterpret_cast<uintptr_t>(dst + i) & (4 * sizeof(TO) - 1)) == 0)) {
            typedef __attribute__((ext_vector_type(4))) TI VI;
            typedef __attribute__((ext_vector_type(4))) TO VO;
            const VI v = *reinterpret_cast<const VI*>(src + i);
            VO o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (TO)(float)v[e];
            *reinterpret_cast<VO*>(dst + i) = o;
        } else {
            for (int64_t j = i; j < n && j < i + 4; ++j) dst[j] = (TO)(float)src[j];
        }
    }
}
}  // namespace

extern "C" int modcr_convert_segments(const void* const* src, void* const* dst, const int64_t* n, int32_t count, int32_t src_dtype,
                                      int32_t dst_dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(src && dst && n && count > 0, "convert_segments: bad arguments");
    MODCR_REQUIRE((src_dtype == MODCR_F32 || src_dtype == MODCR_BF16) && (dst_dtype == MODCR_F32 || dst_dtype == MODCR_BF16),
                  "convert_segments: fp32 / bf16 only");
    hipStream_t st = (hipStream_t)stream;
    for (int32_t base = 0; base < count; base += 8) {
        ConvSegs sg;
        int64_t longest = 0;
        const int m = count - base < 8 ? count - base : 8;
        for (int k = 0; k < 8; ++k) {
            const int q = k < m ? k : 0;                      // unused slots repeat slot 0 with n = 0
            MODCR_REQUIRE(src[base + q] && dst[base + q] && n[base + q] >= 0, "convert_segments: null segment");
            sg.src[k] = src[base + q]; sg.dst[k] = dst[base + q]; sg.n[k] = k < m ? n[base + q] : 0;
            if (sg.n[k] > longest) longest = sg.n[k];
        }
        if (!longest) continue;
        int64_t bx = (longest + 1023) / 1024;
        if (bx > 2048) bx = 2048;
        const dim3 grid((unsigned)bx, (unsigned)m), blk(256);
        if (src_dtype == MODCR_F32 && dst_dtype == MODCR_BF16) hipLaunchKernelGGL((convert_segments_kernel<float, bf16>), grid, blk, 0, st, sg);
        else if (src_dtype == MODCR_BF16 && dst_dtype == MODCR_F32) hipLaunchKernelGGL((convert_segments_kernel<bf16, float>), grid, blk, 0, st, sg);
        else if (src_dtype == MODCR_F32) hipLaunchKernelGGL((convert_segments_kernel<float, float>), grid, blk, 0, st, sg);
        else hipLaunchKernelGGL((convert_segments_kernel<bf16, bf16>), grid, blk, 0, st, sg);
        const int rc = modcr_check_launch("convert_segments");
        if (rc != MODCR_OK) return rc;
    }
    return MODCR_OK;
}

extern "C" int modcr_split3_bf16(const float* src, int64_t lds_, void* dst, int64_t ldd, int64_t M, int32_t K,
                                 int32_t mode, modcr_stream_t stream) {
    MODCR_REQUIRE(src && dst && M > 0 && K > 0 && lds_ >= K && ldd >= 3 * (int64_t)K && (mode == 0 || mode == 1),
                  "split3_bf16: bad arguments");
    hipLaunchKernelGGL(split3_kernel, dim3(blocks_for(M * K, 256)), dim3(256), 0, (hipStream_t)stream, src, lds_,
                       (bf16*)dst, ldd, M, K, mode);
    return modcr_check_launch("split3_bf16");
}

extern "C" int modcr_pack_mask_bits(const float* mask, uint32_t* bits, int64_t rows, int32_t L,
                                    modcr_stream_t stream) {
    MODCR_REQUIRE(mask && bits && rows > 0 && L > 0, "pack_mask_bits: bad arguments");
    const int LW = (L + 31) / 32;
    hipLaunchKernelGGL(pack_bits_kernel, dim3(blocks_for(rows * LW, 256)), dim3(256), 0, (hipStream_t)stream,
                       mask, bits, rows, L, LW);
    return modcr_check_launch("pack_mask_bits");
}

extern "C" int modcr_build_packed_mask(const float* key_mask, uint32_t* bits, int32_t N, int32_t S, int32_t k, modcr_stream_t stream) {
    MODCR_REQUIRE(key_mask && bits && N > 0 && S > 0 && k > 0 && (N % k) == 0, "build_packed_mask: bad arguments (N=%d S=%d k=%d)", N, S, k);
    const int64_t nb = N / k;
    const int L = S * k, LW = (L + 31) / 32;
    hipLaunchKernelGGL(packed_mask_kernel, dim3(blocks_for(nb * L * LW, 256)), dim3(256), 0, (hipStream_t)stream, key_mask, bits, nb, S, k);
    return modcr_check_launch("build_packed_mask");
}

extern "C" int modcr_build_phase_mask(const float* input_mask, const float* chunk_mask, uint32_t* bits,
                                      int32_t N, int32_t T, int32_t R, int32_t phase, modcr_stream_t stream) {
    MODCR_REQUIRE(input_mask && chunk_mask && bits && N > 0 && T > 0 && R >= 0, "build_phase_mask: bad arguments");
    MODCR_REQUIRE(phase == 1 || phase == 3, "build_phase_mask: phase %d (phase 2 is the broadcast key mask)", phase);
    const int S = T + R, LW = (S + 31) / 32;
    hipLaunchKernelGGL(phase_mask_kernel, dim3(blocks_for((int64_t)N * S * LW, 256)), dim3(256), 0,
                       (hipStream_t)stream, input_mask, chunk_mask, bits, N, T, R, phase);
    return modcr_check_launch("build_phase_mask");
}

extern "C" int modcr_chunk_mean_q_fwd(void* q, int64_t row_stride, int64_t seq_stride,
                                      const int32_t* chunk_id, int32_t N, int32_t T, int32_t H,
                                      int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(q && chunk_id && N > 0 && T > 0 && H > 0, "chunk_mean_q_fwd: bad arguments");
    const dim3 grid((H + 63) / 64, N), blk(256);
    const size_t shm = (3 * (size_t)((T + 3) & ~3) + (size_t)T * 64) * sizeof(float);
    MODCR_REQUIRE(shm <= 160 * 1024, "chunk_mean_q_fwd: T=%d too long", T);
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];         // write-once: more than the default 64 KB of LDS for T > 238
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_mean_q_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chunk_mean_q_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        configured = true;
    }
    if (dtype == MODCR_BF16)
        hipLaunchKernelGGL((chunk_mean_q_kernel<bf16>), grid, blk, shm, (hipStream_t)stream, (bf16*)q, row_stride, seq_stride, chunk_id, T, H);
    else
        hipLaunchKernelGGL((chunk_mean_q_kernel<float>), grid, blk, shm, (hipStream_t)stream, (float*)q, row_stride, seq_stride, chunk_id, T, H);
    return modcr_check_launch("chunk_mean_q");
}

// shapes the row-piece kernels take: 8-feature chunks, whole heads per chunk group, <= 8 heads (32 lanes each
// in the softmax), at least one key slot per 256 threads, 16-byte aligned rows
// LDS bytes of the alignment-attention kernels and the key block LB of their score phase: [max(LB * C, KS * E)] partial
// products + tables x [L][heads]; LB = all keys when they fit in ~144 KB, else the largest multiple of 4 KS that does
static size_t align_attn_lds(int L, int E, int heads, int tables, int* LB) {
    const int C = E / 8, KS = 256 / C;
    const size_t budget = 144 * 1024, tab = (size_t)tables * L * heads * sizeof(float), out = (size_t)KS * E * sizeof(float);
    if (tab + out > budget) return 0;
    int lb = (int)((budget - tab) / ((size_t)C * sizeof(float)));
    if (lb >= L) lb = L;
    else lb = lb / (4 * KS) * (4 * KS);
    if (lb < 4 * KS && lb < L) return 0;
    *LB = lb;
    const size_t part = (size_t)lb * C * sizeof(float) > out ? (size_t)lb * C * sizeof(float) : out;
    return part + tab;
}

static bool align_attn_shape_ok(const void* k, const void* v, int64_t ldkv, int L, int E, int heads, int dtype) {
    const int C = E / 8;
    const int esz = dtype == MODCR_BF16 ? 2 : 4;
    return (E % 8) == 0 && C <= 256 && (C % heads) == 0 && heads <= 8 && (ldkv * esz) % 16 == 0 &&
           modcr_aligned16(k) && modcr_aligned16(v);
}

extern "C" int modcr_align_attn_fwd(const float* q, const void* k, const void* v, int64_t ldkv, float* out,
                                    float* probs, int32_t N, int32_t L, int32_t E, int32_t heads, float scale,
                                    float p, uint64_t seed, uint64_t offset, const float* key_bias, int32_t dtype,
                                    modcr_stream_t stream) {
    MODCR_REQUIRE(q && k && v && out, "align_attn_fwd: null pointer");
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "align_attn_fwd: p=%g out of [0, 1)", p);
    const uint32_t thr = p > 0.f ? (uint32_t)((double)p * 32768.0 + 0.5) : 0u;      // 15-bit uniforms (drop_field)
    const float keep_scale = 1.0f / (1.0f - p);
    MODCR_REQUIRE(N > 0 && L > 0 && heads > 0 && E % heads == 0 && ldkv >= E, "align_attn_fwd: bad shape");
    MODCR_REQUIRE(align_attn_shape_ok(k, v, ldkv, L, E, heads, dtype),
                  "align_attn_fwd: needs E %% 8 == 0, (E/8) %% heads == 0, heads <= 8, E <= 2048, 16-byte aligned rows (E=%d heads=%d)", E, heads);
    int LB;
    const size_t shm = align_attn_lds(L, E, heads, 1, &LB);
    MODCR_REQUIRE(shm && shm <= 160 * 1024, "align_attn_fwd: L=%d too long", L);
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&align_attn_fwd_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&align_attn_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        configured = true;
    }
    const dim3 grid(N), blk(256);
    if (dtype == MODCR_BF16)
        hipLaunchKernelGGL((align_attn_fwd_kernel<bf16>), grid, blk, shm, (hipStream_t)stream, q,
                           (const bf16*)k, (const bf16*)v, ldkv, out, probs, L, E, heads, scale, seed, offset, thr, keep_scale, key_bias, LB);
    else
        hipLaunchKernelGGL((align_attn_fwd_kernel<float>), grid, blk, shm, (hipStream_t)stream, q,
                           (const float*)k, (const float*)v, ldkv, out, probs, L, E, heads, scale, seed, offset, thr, keep_scale, key_bias, LB);
    return modcr_check_launch("align_attn_fwd");
}

extern "C" int modcr_align_attn_bwd(const float* dout, const float* q, const void* k, const void* v,
                                    int64_t ldkv, const float* probs, float* dq, void* dk, void* dv,
                                    int64_t lddkv, int32_t N, int32_t L, int32_t E, int32_t heads, float scale,
                                    float p, uint64_t seed, uint64_t offset, int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(dout && q && k && v && probs && dq && dk && dv, "align_attn_bwd: null pointer");
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "align_attn_bwd: p=%g out of [0, 1)", p);
    const uint32_t thr = p > 0.f ? (uint32_t)((double)p * 32768.0 + 0.5) : 0u;      // 15-bit uniforms (drop_field)
    const float keep_scale = 1.0f / (1.0f - p);
    MODCR_REQUIRE(N > 0 && L > 0 && heads > 0 && E % heads == 0 && ldkv >= E && lddkv >= E, "align_attn_bwd: bad shape");
    MODCR_REQUIRE(align_attn_shape_ok(k, v, ldkv, L, E, heads, dtype) && align_attn_shape_ok(dk, dv, lddkv, L, E, heads, dtype),
                  "align_attn_bwd: needs E %% 8 == 0, (E/8) %% heads == 0, heads <= 8, E <= 2048, 16-byte aligned rows (E=%d heads=%d)", E, heads);
    int LB;
    const size_t shm = align_attn_lds(L, E, heads, 2, &LB);
    MODCR_REQUIRE(shm && shm <= 160 * 1024, "align_attn_bwd: L=%d too long", L);
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&align_attn_bwd_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&align_attn_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        configured = true;
    }
    const dim3 grid(N), blk(256);
    if (dtype == MODCR_BF16)
        hipLaunchKernelGGL((align_attn_bwd_kernel<bf16>), grid, blk, shm, (hipStream_t)stream, dout, q,
                           (const bf16*)k, (const bf16*)v, ldkv, probs, dq, (bf16*)dk, (bf16*)dv, lddkv, L, E,
                           heads, scale, seed, offset, thr, keep_scale, LB);
    else
        hipLaunchKernelGGL((align_attn_bwd_kernel<float>), grid, blk, shm, (hipStream_t)stream, dout, q,
                           (const float*)k, (const float*)v, ldkv, probs, dq, (float*)dk, (float*)dv, lddkv, L, E,
                           heads, scale, seed, offset, thr, keep_scale, LB);
    return modcr_check_launch("align_attn_bwd");
}

extern "C" int modcr_mc_ce_fwd_bwd(const float* logits, const float* label, float* loss, float* dlogits,
                                   const float* grad_scale, int32_t B, int32_t C, modcr_stream_t stream) {
    MODCR_REQUIRE(logits && label && (loss || dlogits) && B > 0 && C > 0, "mc_ce_fwd_bwd: bad arguments");
    hipLaunchKernelGGL(mc_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, label, loss, dlogits,
                       grad_scale, B, C);
    return modcr_check_launch("mc_ce");
}

extern "C" int modcr_layernorm_bwd(const float* dY, const float* x, const float* residual, const float* gamma,
                                   float eps, float* dX, float* dgamma, float* dbeta, int64_t M, int32_t H,
                                   modcr_stream_t stream) {
    MODCR_REQUIRE(dY && x && gamma && dX && M > 0 && H > 0, "layernorm_bwd: bad arguments");
    if (H <= 1024) {
        // rows per wave: enough blocks to fill the chip (>= 1024 where M allows), at most 16 rows per wave
        int rpw = (int)(M / (4 * 1024));
        rpw = rpw < 1 ? 1 : (rpw > 16 ? 16 : rpw);
        const dim3 grid(blocks_for(M, 4 * rpw));
        if (H <= 256) hipLaunchKernelGGL(layernorm_bwd_rows_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dY, x, residual, gamma, eps, dX, dgamma, dbeta, M, H, rpw);
        else if (H <= 768) hipLaunchKernelGGL(layernorm_bwd_rows_kernel<12>, grid, dim3(256), 0, (hipStream_t)stream, dY, x, residual, gamma, eps, dX, dgamma, dbeta, M, H, rpw);
        else hipLaunchKernelGGL(layernorm_bwd_rows_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, dY, x, residual, gamma, eps, dX, dgamma, dbeta, M, H, rpw);
        return modcr_check_launch("layernorm_bwd");
    }
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks_for(M, 4)), dim3(256), 0, (hipStream_t)stream, dY, x,
                       residual, gamma, eps, dX, dgamma, dbeta, M, H);
    return modcr_check_launch("layernorm_bwd");
}

extern "C" int modcr_act_bwd(const float* dact, const float* pre, float* dpre, int64_t n, int32_t act,
                             modcr_stream_t stream) {
    MODCR_REQUIRE(dact && pre && dpre && n > 0, "act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dact, pre, dpre, n, act);
    return modcr_check_launch("act_bwd");
}

// ---- optimizer step over flat fp32 buffers (HBM-bound: 16 B read + 12 B written per parameter) -------------
namespace {
__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, int64_t n, float* out) {
    __shared__ float part[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += x[i] * x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// The same sum with a FIXED summation order: block b writes its partial to partials[b], the last launch folds them in index order.
// The clip coefficient of a data-parallel step must come out bit-identical on every rank (all ranks hold the same reduced
// gradient; a float atomicAdd from 2048 workgroups lands in whatever order the hardware schedules them, and replicas whose
// clip differs in the last bit drift apart one ulp per step).
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* x, int64_t n, float* partials) {
    __shared__ float part[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) acc += x[i] * x[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* partials, int count, float* out) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) acc += partials[i];          // thread t: partials t, t + 256, ... in order
    acc = wave_sum(acc);                                                        // fixed butterfly
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out += (part[0] + part[1]) + (part[2] + part[3]);
}

// HF = 1: transformers.AdamW (the optimizer the reference trains with, run_PMR_ModCR.py:24,137; transformers 4.x
// optimization.py, correct_bias=True): denom = sqrt(v) + eps, step = lr * sqrt(bc2) / bc1, decoupled decay applied AFTER
// the update (p -= lr * wd * p).  HF = 0: torch.optim.AdamW (decay first, denom = sqrt(v) / sqrt(bc2) + eps).  The two
// differ in where eps enters: with eps = 1e-5 and clipped gradients of ~1e-4 per element that is not a rounding matter.
template <int HF>
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                   const float* sumsq, float max_norm, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2) {
    float clip = 1.0f;
    if (sumsq && max_norm > 0.f) clip = fminf(1.0f, max_norm / (sqrtf(*sumsq) + 1e-6f));
    const float step = HF ? lr * sqrtf(bc2) / bc1 : lr / bc1, rs2 = 1.0f / sqrtf(bc2), decay = 1.0f - lr * wd;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= clip;
        if (!HF) pp *= decay;
        mm = b1 * mm + (1.0f - b1) * gg;
        vv = b2 * vv + (1.0f - b2) * gg * gg;
        if (HF) { pp -= step * mm / (sqrtf(vv) + eps); pp *= decay; }
        else pp -= step * mm / (sqrtf(vv) * rs2 + eps);
    };
    // the four buffers share one misalignment (slices of flat buffers at the same offset): scalar head up to the
    // first 16-byte boundary, 16-byte body, scalar tail
    int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15) >> 2);
    if (head > n) head = n;
    if (blockIdx.x == 0)
        for (int64_t i = threadIdx.x; i < head; i += 256) upd(p[i], g[i], m[i], v[i]);
    p += head; g += head; m += head; v += head; n -= head;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i], mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pp[e], me = mm[e], ve = vv[e];
            upd(pe, gg[e], me, ve);
            pp[e] = pe; mm[e] = me; vv[e] = ve;
        }
        reinterpret_cast<f32x4*>(p)[i] = pp; reinterpret_cast<f32x4*>(m)[i] = mm; reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
}
}  // namespace

extern "C" int modcr_sumsq_f32(const float* x, int64_t n, float* out, modcr_stream_t stream) {
    MODCR_REQUIRE(x && out && n > 0, "sumsq_f32: bad arguments");
    MODCR_REQUIRE(modcr_aligned16(x), "sumsq_f32: 16-byte alignment");
    const int grid = (int)((n / 4 + 255) / 256 < 2048 ? ((n / 4 + 255) / 256 > 0 ? (n / 4 + 255) / 256 : 1) : 2048);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return modcr_check_launch("sumsq_f32");
}

extern "C" int modcr_sumsq_partials(void) { return 2048; }

extern "C" int modcr_sumsq_f32_ordered(const float* x, int64_t n, float* out, float* partials, int32_t max_partials, modcr_stream_t stream) {
    MODCR_REQUIRE(x && out && partials && n > 0 && max_partials > 0, "sumsq_f32_ordered: bad arguments");
    MODCR_REQUIRE(modcr_aligned16(x), "sumsq_f32_ordered: 16-byte alignment");
    int64_t grid = (n / 4 + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    if (grid > max_partials) grid = max_partials;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, x, n, partials);
    hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)grid, out);
    return modcr_check_launch("sumsq_f32_ordered");
}

static int adamw_launch(int hf, float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq,
                        float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                        float bc1, float bc2, modcr_stream_t stream) {
    MODCR_REQUIRE(p && g && m && v && n > 0, "adamw_step: bad arguments");
    MODCR_REQUIRE(((uintptr_t)p & 3) == 0 && ((uintptr_t)p & 15) == ((uintptr_t)g & 15) && ((uintptr_t)p & 15) == ((uintptr_t)m & 15) &&
                      ((uintptr_t)p & 15) == ((uintptr_t)v & 15), "adamw_step: p, g, m, v must share one alignment modulo 16 bytes");
    MODCR_REQUIRE(bc1 > 0.f && bc2 > 0.f, "adamw_step: bias corrections must be positive");
    const int grid = (int)((n / 4 + 255) / 256 < 4096 ? ((n / 4 + 255) / 256 > 0 ? (n / 4 + 255) / 256 : 1) : 4096);
    if (hf) hipLaunchKernelGGL(adamw_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, max_norm, lr,
                               beta1, beta2, eps, weight_decay, bc1, bc2);
    else hipLaunchKernelGGL(adamw_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, max_norm, lr,
                            beta1, beta2, eps, weight_decay, bc1, bc2);
    return modcr_check_launch("adamw_step");
}

extern "C" int modcr_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq,
                                float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                float bc1, float bc2, modcr_stream_t stream) {
    return adamw_launch(0, p, g, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2, stream);
}

extern "C" int modcr_adamw_hf_step(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq,
                                   float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                   float bc1, float bc2, modcr_stream_t stream) {
    return adamw_launch(1, p, g, m, v, n, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, bc1, bc2, stream);
}


// ---- out = a + b (residual-gradient sums of the layer backward): a fp32, b fp32 or bf16, out fp32 or bf16 --------
namespace {
template <typename TB, typename TO>
__global__ __launch_bounds__(256) void add_kernel(const float* a, const TB* b, TO* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = from_f32<TO>(a[i] + to_f32(b[i]));
}
}  // namespace

extern "C" int modcr_add(const float* a, const void* b, int32_t b_dtype, void* out, int32_t out_dtype, int64_t n,
                         modcr_stream_t stream) {
    MODCR_REQUIRE(a && b && out && n > 0, "add: bad arguments");
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipStream_t st = (hipStream_t)stream;
    const int key = b_dtype * 2 + out_dtype;
    if (key == MODCR_F32 * 2 + MODCR_F32) hipLaunchKernelGGL((add_kernel<float, float>), dim3(grid), dim3(256), 0, st, a, (const float*)b, (float*)out, n);
    else if (key == MODCR_F32 * 2 + MODCR_BF16) hipLaunchKernelGGL((add_kernel<float, bf16>), dim3(grid), dim3(256), 0, st, a, (const float*)b, (bf16*)out, n);
    else if (key == MODCR_BF16 * 2 + MODCR_F32) hipLaunchKernelGGL((add_kernel<bf16, float>), dim3(grid), dim3(256), 0, st, a, (const bf16*)b, (float*)out, n);
    else if (key == MODCR_BF16 * 2 + MODCR_BF16) hipLaunchKernelGGL((add_kernel<bf16, bf16>), dim3(grid), dim3(256), 0, st, a, (const bf16*)b, (bf16*)out, n);
    else { modcr_set_error("add: unknown dtypes %d / %d", b_dtype, out_dtype); return MODCR_ERR_INVALID; }
    return modcr_check_launch("add");
}


// adjoint of the chunk-mean query (v10:66-78): every row of a chunk receives the mean of the chunk's gradient rows --
// the same segment mean applied to dq
extern "C" int modcr_chunk_mean_q_bwd(void* dq, int64_t row_stride, int64_t seq_stride, const int32_t* chunk_id,
                                      int32_t N, int32_t T, int32_t H, int32_t dtype, modcr_stream_t stream) {
    return modcr_chunk_mean_q_fwd(dq, row_stride, seq_stride, chunk_id, N, T, H, dtype, stream);
}


// ---- dropout (train-mode semantics of the reference: nn.Dropout inside the frozen encoders stays active under
// model.train(), run_PMR_ModCR.py:171).  Counter-based: element i of a call keeps its value iff
// hash(seed, offset + i) >= p, so the backward pass (and any recomputation) regenerates the mask from (seed, offset)
// instead of storing it.  hash: drop_words above (one finaliser + one multiply-xorshift per group of four counters, 15-bit uniforms).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* x, T* out, int64_t n, uint64_t seed, uint64_t offset,
                                                     uint32_t thr, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = from_f32<T>(drop_apply(to_f32(x[i]), seed, offset + (uint64_t)i, thr, scale));
}

// y = LN(dropout(x) + residual): BertSelfOutput / BertOutput in training mode (a_bert:369-373, :446-451)
template <typename TX, typename TR, typename TO>
__global__ __launch_bounds__(256) void layernorm_dropout_kernel(const TX* x, const TR* res, const float* gamma,
                                                                const float* beta, float eps, TO* y, void* pre_out, int pre_f16, int64_t M, int H,
                                                                uint64_t seed, uint64_t offset, uint32_t thr, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int nv = (H + 255) / 256;
    float v[MAXV][4];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (i < nv && c < H) {
            Vec4<TX>::load(x + m * H + c, v[i]);
            float r[4] = {0.f, 0.f, 0.f, 0.f};
            if (res) Vec4<TR>::load(res + m * H + c, r);
            if (thr) drop_apply4(v[i], seed, offset + (uint64_t)(m * H + c), thr, scale);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] += r[j];
            // the pre-LayerNorm rows, for a backward pass that wants them (trainable layers)
            // (fp32, or IEEE half: 11 significant bits against the 8 of the bf16 activations saved beside them)
            if (pre_out) {
                if (pre_f16) Vec4<_Float16>::store(reinterpret_cast<_Float16*>(pre_out) + m * H + c, v[i]);
                else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(pre_out) + m * H + c) = f32x4{v[i][0], v[i][1], v[i][2], v[i][3]};
            }
        }
    }
    ln_finish<TO>(v, nv, H, lane, gamma, beta, eps, y + m * H);
}

// LayerNorm backward, 16-byte pieces (H = 256 NV: a lane owns columns 4 lane + 256 k .. + 3), dY fp32 or bf16.
// Writes the fp32 gradient of the pre-LN rows (= residual branch) and, optionally, a bf16 copy with the dropout
// mask of the forward applied (= gradient of the GEMM output of BertSelfOutput / BertOutput): the operand of the
// two backward GEMMs leaves this kernel in the dtype and with the mask they need, no separate dropout / cast pass.
template <int NV, typename TDY, typename TP>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const TDY* dY, const TP* pre, const float* gamma, float eps,
                                                                float* dX, bf16* dXb, float* dgamma, float* dbeta, int64_t M,
                                                                int rpw, uint64_t seed, uint64_t offset, uint32_t thr, float scale) {
    constexpr int H = 256 * NV;
    __shared__ float sPart[2][4][H];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pg[NV][4], pb[NV][4], gm[NV][4];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        Vec4<float>::load(gamma + 4 * lane + 256 * k, gm[k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pg[k][j] = 0.f; pb[k][j] = 0.f; }
    }
    const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * rpw;
    for (int64_t m = m0; m < m0 + rpw && m < M; ++m) {
        float xr[NV][4], dy[NV][4];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int64_t at = m * H + 4 * lane + 256 * k;
            Vec4<TP>::load(pre + at, xr[k]);
            Vec4<TDY>::load(dY + at, dy[k]);
            s += xr[k][0] + xr[k][1] + xr[k][2] + xr[k][3];
        }
        const float mean = wave_sum(s) * (1.0f / H);
        float qv = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = xr[k][j] - mean; qv += d * d; }
        const float rstd = rsqrtf(wave_sum(qv) * (1.0f / H) + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xr[k][j] = (xr[k][j] - mean) * rstd;
                const float g = dy[k][j] * gm[k][j];
                a += g; b += g * xr[k][j];
            }
        a = wave_sum(a) * (1.0f / H);
        b = wave_sum(b) * (1.0f / H);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int64_t at = m * H + 4 * lane + 256 * k;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = rstd * (dy[k][j] * gm[k][j] - a - xr[k][j] * b);
                pg[k][j] += dy[k][j] * xr[k][j];
                pb[k][j] += dy[k][j];
            }
            if (dX) Vec4<float>::store(dX + at, o);
            if (dXb) {
                if (thr) drop_apply4(o, seed, offset + (uint64_t)at, thr, scale);
                Vec4<bf16>::store(dXb + at, o);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) { sPart[0][wave][4 * lane + 256 * k + j] = pg[k][j]; sPart[1][wave][4 * lane + 256 * k + j] = pb[k][j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        if (dgamma) atomicAdd(dgamma + c, sPart[0][0][c] + sPart[0][1][c] + sPart[0][2][c] + sPart[0][3][c]);
        if (dbeta) atomicAdd(dbeta + c, sPart[1][0][c] + sPart[1][1][c] + sPart[1][2][c] + sPart[1][3][c]);
    }
}
}  // namespace

extern "C" int modcr_dropout(const void* x, void* out, int64_t n, int32_t dtype, float p, uint64_t seed, uint64_t offset,
                             modcr_stream_t stream) {
    MODCR_REQUIRE(x && out && n > 0, "dropout: bad arguments");
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "dropout: p=%g out of [0, 1)", p);
    const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    const uint32_t thr = drop_threshold(p);
    const float scale = 1.0f / (1.0f - p);
    if (dtype == MODCR_BF16)
        hipLaunchKernelGGL((dropout_kernel<bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, n, seed, offset, thr, scale);
    else
        hipLaunchKernelGGL((dropout_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)out, n, seed, offset, thr, scale);
    return modcr_check_launch("dropout");
}

// Rows [M, H] -> rows row0 + (m % rpg) of sequence m / rpg of a [*, gstride, H] buffer, with nn.Dropout on the way (p = 0: a
// plain strided copy): the LayerNorm-ed region rows of modeling_transfomres.py:676-684 / v10:338-345 go behind the text rows of
// each sequence (the torch.cat) under img dropout in ONE pass -- round 1-4 copied them and then ran modcr_dropout over the
// whole buffer (27 + 44 us at S = 101, 50 + 91 us at S = 180, three encoder passes per step).  Counter of element (dst row, c)
// = offset + dst_row * H + c: the flat index of the destination buffer, the mask modcr_dropout over that buffer applies.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void rows_scatter_dropout_kernel(const T* src, T* dst, int64_t M, int H, int rpg, int64_t gstride, int row0,
                                                                  uint64_t seed, uint64_t offset, uint32_t thr, float scale) {
    const int pieces = H >> 2;                                       // 4 elements per thread and step
    const int64_t total = M * pieces;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / pieces;
        const int c = (int)(i - m * pieces) * 4;
        const int64_t drow = (m / rpg) * gstride + row0 + (m % rpg);
        float v[4];
        Vec4<T>::load(src + m * H + c, v);
        if (thr) drop_apply4(v, seed, offset + (uint64_t)(drow * H + c), thr, scale);
        Vec4<T>::store(dst + drow * H + c, v);
    }
}
}  // namespace

extern "C" int modcr_rows_scatter_dropout(const void* src, void* dst, int64_t M, int32_t H, int32_t rows_per_group, int64_t group_stride,
                                          int32_t row0, int32_t dtype, float p, uint64_t seed, uint64_t offset, modcr_stream_t stream) {
    MODCR_REQUIRE(src && dst && M > 0 && H > 0 && (H % 4) == 0, "rows_scatter_dropout: bad arguments");
    MODCR_REQUIRE(rows_per_group > 0 && row0 >= 0 && group_stride >= (int64_t)row0 + rows_per_group,
                  "rows_scatter_dropout: rows_per_group=%d + row0=%d do not fit group_stride=%lld", rows_per_group, row0, (long long)group_stride);
    MODCR_REQUIRE(dtype == MODCR_BF16 || dtype == MODCR_F32, "rows_scatter_dropout: dtype");
    {   // four elements per access: 8-byte (bf16) / 16-byte (fp32) aligned rows on both sides
        const uintptr_t am = dtype == MODCR_BF16 ? 7 : 15;
        MODCR_REQUIRE(((uintptr_t)src & am) == 0 && ((uintptr_t)dst & am) == 0, "rows_scatter_dropout: src / dst must be %d-byte aligned", (int)am + 1);
    }
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "rows_scatter_dropout: p=%g out of [0, 1)", p);
    const int64_t total = M * (H / 4);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    const uint32_t thr = p > 0.f ? drop_threshold(p) : 0u;
    const float scale = 1.0f / (1.0f - p);
    if (dtype == MODCR_BF16)
        hipLaunchKernelGGL((rows_scatter_dropout_kernel<bf16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, (bf16*)dst, M, H,
                           rows_per_group, group_stride, row0, seed, offset, thr, scale);
    else
        hipLaunchKernelGGL((rows_scatter_dropout_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst, M, H,
                           rows_per_group, group_stride, row0, seed, offset, thr, scale);
    return modcr_check_launch("rows_scatter_dropout");
}

extern "C" int modcr_dropout_residual_ln_fwd(const void* x, int32_t x_dtype, const void* residual, int32_t res_dtype, const float* gamma,
                                             const float* beta, float eps, void* out, int32_t out_dtype, void* pre_out, int32_t pre_dtype,
                                             int64_t M, int32_t H, float p, uint64_t seed, uint64_t offset, modcr_stream_t stream) {
    MODCR_REQUIRE(x && gamma && beta && out && M > 0 && (H % 4) == 0 && H <= 256 * MAXV, "dropout_residual_ln_fwd: bad arguments");
    MODCR_REQUIRE(!pre_out || pre_dtype == MODCR_F32 || pre_dtype == MODCR_F16, "dropout_residual_ln_fwd: pre_out is fp32 or IEEE half");
    const int pre_f16 = pre_dtype == MODCR_F16;
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "dropout_residual_ln_fwd: p=%g out of [0, 1)", p);
    const dim3 grid(blocks_for(M, 4)), blk(256);
    const uint32_t thr = drop_threshold(p);
    const float scale = 1.0f / (1.0f - p);
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == MODCR_F16) {             // fp16 sublayer output of the bf16 path
        MODCR_REQUIRE((!residual || res_dtype == MODCR_BF16) && out_dtype == MODCR_BF16, "dropout_residual_ln_fwd: fp16 rows need a bf16 residual and output");
        hipLaunchKernelGGL((layernorm_dropout_kernel<_Float16, bf16, bf16>), grid, blk, 0, st, (const _Float16*)x, (const bf16*)residual, gamma,
                           beta, eps, (bf16*)out, pre_out, pre_f16, M, H, seed, offset, thr, scale);
        return modcr_check_launch("dropout_residual_ln");
    }
    MODCR_REQUIRE(x_dtype == MODCR_F32, "dropout_residual_ln_fwd: x must be fp32 or fp16");
    const int key = (residual ? res_dtype : MODCR_F32) * 2 + out_dtype;
#define LND_CASE(K, TR, TO)                                                                                         \
    case K:                                                                                                        \
        hipLaunchKernelGGL((layernorm_dropout_kernel<float, TR, TO>), grid, blk, 0, st, (const float*)x, (const TR*)residual, gamma, beta, eps, \
                           (TO*)out, pre_out, pre_f16, M, H, seed, offset, thr, scale);                            \
        break;
    switch (key) {
        LND_CASE(0, bf16, bf16) LND_CASE(1, bf16, float) LND_CASE(2, float, bf16) LND_CASE(3, float, float)
        default: MODCR_REQUIRE(false, "dropout_residual_ln_fwd: bad dtypes");
    }
#undef LND_CASE
    return modcr_check_launch("dropout_residual_ln");
}


// LayerNorm backward for the encoder layer's two output blocks: dY fp32 or bf16 -> d_pre fp32 (may be NULL) and / or a
// bf16 copy with the forward's dropout mask (p, seed, offset; p = 0: plain copy).  H in {256, 512, 768, 1024}.
extern "C" int modcr_layernorm_dropout_bwd(const void* dY, int32_t dy_dtype, const void* pre, int32_t pre_dtype, const float* gamma, float eps,
                                           float* d_pre, void* d_sub_bf16, float* dgamma, float* dbeta, int64_t M, int32_t H,
                                           float p, uint64_t seed, uint64_t offset, modcr_stream_t stream) {
    MODCR_REQUIRE(pre_dtype == MODCR_F32 || pre_dtype == MODCR_F16, "layernorm_dropout_bwd: the pre-LayerNorm rows are fp32 or IEEE half");
    MODCR_REQUIRE(dY && pre && gamma && (d_pre || d_sub_bf16) && M > 0, "layernorm_dropout_bwd: bad arguments");
    MODCR_REQUIRE(H % 256 == 0 && H <= 1024, "layernorm_dropout_bwd: H=%d must be 256, 512, 768 or 1024", H);
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "layernorm_dropout_bwd: p=%g out of [0, 1)", p);
    MODCR_REQUIRE(dy_dtype == MODCR_F32 || dy_dtype == MODCR_BF16, "layernorm_dropout_bwd: dY dtype");
    int rpw = (int)(M / (4 * 1024));
    rpw = rpw < 1 ? 1 : (rpw > 16 ? 16 : rpw);
    const dim3 grid(blocks_for(M, 4 * rpw));
    const uint32_t thr = p > 0.f ? drop_threshold(p) : 0u;
    const float scale = 1.0f / (1.0f - p);
    hipStream_t st = (hipStream_t)stream;
#define MODCR_LNB(NV, T, TP) hipLaunchKernelGGL((layernorm_bwd_vec_kernel<NV, T, TP>), grid, dim3(256), 0, st, (const T*)dY, (const TP*)pre, gamma, eps, \
                                                d_pre, (bf16*)d_sub_bf16, dgamma, dbeta, M, rpw, seed, offset, thr, scale)
#define MODCR_LNB_H(T, TP) switch (H / 256) { case 1: MODCR_LNB(1, T, TP); break; case 2: MODCR_LNB(2, T, TP); break; case 3: MODCR_LNB(3, T, TP); break; default: MODCR_LNB(4, T, TP); }
    if (pre_dtype == MODCR_F16) {
        if (dy_dtype == MODCR_F32) { MODCR_LNB_H(float, _Float16) } else { MODCR_LNB_H(bf16, _Float16) }
    } else {
        if (dy_dtype == MODCR_F32) { MODCR_LNB_H(float, float) } else { MODCR_LNB_H(bf16, float) }
    }
#undef MODCR_LNB_H
#undef MODCR_LNB
    return modcr_check_launch("layernorm_dropout_bwd");
}


// ---- embedding-table backward (autograd of a_transformers.../modeling_bert.py:184-211's three lookups; VERDICT r03 "missing" 5):
// dW[id] += sum of the gradient rows that looked `id` up.  The caller sorts the flat ids (stable) and hands the permutation:
// workgroup b of the sorted order is the OWNER of the segment that starts at b (every other workgroup returns at once), walks
// its rows in sorted order and adds the sum to its table row -- one writer per row, a fixed summation order: deterministic,
// no atomics.  Rows of padding_idx are skipped (nn.Embedding(padding_idx=) leaves that row without gradient).  A segment is
// read by ONE workgroup: fine for word ids (longest real segments: [CLS] / [SEP], N .. 2N rows) and position ids (N rows each);
// tables of a handful of rows (token types) go through two row reductions instead (modcr_hip.embedding_bwd).
namespace {
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ sid, const int64_t* __restrict__ order,
                                                            const float* __restrict__ dy, float* __restrict__ dw, int M, int H, int64_t pad,
                                                            int64_t V) {
    const int b = blockIdx.x;
    const int64_t id = sid[b];
    // (an id outside the table cannot come from a forward that ran -- the lookup would have faulted -- but a caller's bug must not
    // turn into a read-modify-write outside dw: such rows are skipped)
    if ((b > 0 && sid[b - 1] == id) || id == pad || id < 0 || id >= V) return;
    int e = b + 1;
    while (e < M && sid[e] == id) ++e;                       // uniform scalar walk (the ids are L2-resident)
    for (int c = 4 * threadIdx.x; c < H; c += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int r = b;
        for (; r + 4 <= e; r += 4) {                          // four independent row loads in flight
            const int64_t o0 = order[r], o1 = order[r + 1], o2 = order[r + 2], o3 = order[r + 3];
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(dy + o0 * H + c), v1 = *reinterpret_cast<const f32x4*>(dy + o1 * H + c);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(dy + o2 * H + c), v3 = *reinterpret_cast<const f32x4*>(dy + o3 * H + c);
            acc += v0; acc += v1; acc += v2; acc += v3;      // (fixed order: r, r+1, r+2, r+3)
        }
        for (; r < e; ++r) acc += *reinterpret_cast<const f32x4*>(dy + order[r] * H + c);
        f32x4* dst = reinterpret_cast<f32x4*>(dw + id * H + c);
        *dst = *dst + acc;
    }
}
}  // namespace

extern "C" int modcr_embedding_bwd_v(const int64_t* sorted_ids, const int64_t* order, const float* dy, float* dw, int32_t M, int32_t H,
                                     int64_t V, int64_t padding_idx, modcr_stream_t stream) {
    MODCR_REQUIRE(sorted_ids && order && dy && dw, "embedding_bwd: null pointer");
    MODCR_REQUIRE(M > 0 && H > 0 && (H % 4) == 0 && V > 0, "embedding_bwd: M = %d, H = %d, V = %lld (H must be a multiple of 4)", M, H, (long long)V);
    MODCR_REQUIRE(modcr_aligned16(dy) && modcr_aligned16(dw), "embedding_bwd: 16-byte alignment of dy / dw");
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, sorted_ids, order, dy, dw, M, H, padding_idx, V);
    return modcr_check_launch("embedding_bwd");
}

// the entry as it was before the table height became an argument (kept at its old signature, so that a caller built against the older
// header still reads `padding_idx` and `stream` where it put them): no upper bound on the ids
extern "C" int modcr_embedding_bwd(const int64_t* sorted_ids, const int64_t* order, const float* dy, float* dw, int32_t M, int32_t H,
                                   int64_t padding_idx, modcr_stream_t stream) {
    return modcr_embedding_bwd_v(sorted_ids, order, dy, dw, M, H, INT64_MAX, padding_idx, stream);
}
