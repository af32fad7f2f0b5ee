// Helpers shared by the attention kernels (attn.hip: forward tile kernels + the older backward cores; attn_bwd.hip: the
// five-product backward core): the attention-probability dropout hash, LDS row swizzles, the backward argument block.
#pragma once
#include "common.h"

namespace {

// nn.Dropout on the softmax output (modeling_bert.py:69 / v10:101): one hash per group of four consecutive keys of a query row
// gives four 15-bit uniforms; a weight whose uniform is below p * 2^15 is zeroed (packed bf16 pairs d0 = keys 4g, 4g+1 and
// d1 = keys 4g+2, 4g+3).  counter = ((n * A + head) * LP + query) * (LP / 4) + key / 4.
// The two hash words of a key group (four 15-bit uniforms: bits 0-14 and 16-30 of each):
//     x = ctr * 0x9E3779B1 ^ s0;  x ^= x >> 15;  x *= 0x85EBCA6B;  x ^= x >> 13;          y = x * 0xC2B2AE35 + s1;  y ^= y >> 16
// Round 3: two 32-bit multiplies per group instead of four (v_mul_lo_u32 is quarter rate: the hash was a third of the VALU time
// of the training-mode kernels) -- the Weyl product ctr * 0x9E3779B1 is a linear function of the counter, so a caller that walks
// counters at a constant stride takes it as `cm` and ADDS stride * 0x9E3779B1 (attn_drop_cm); keep fraction, pairwise independence
// of the four fields / of neighbouring key groups, queries, heads and sequences, and field histograms were compared with the
// round-2 form (two finaliser rounds) on 8 M decisions x 5 seeds: indistinguishable.
// Round 4, measured and NOT adopted: the same hash on two full-rate 24-bit multiply-adds (v_mad_u32_u24) + one more xorshift
// instead of the two v_mul_lo_u32 -- statistically equivalent (tools/hash_stats.py, variant r4c), and in a same-process A/B of the
// two builds (tools/ab_drop_hash.py) not faster: 394.7 vs 391.6 us at N = 512 (eval mode 363.9): the multiplies are not what the
// masking costs; a bench-line difference between two devices had suggested otherwise.
#define MODCR_DROP_WEYL 0x9E3779B1u
__device__ __forceinline__ uint32_t attn_drop_cm(uint32_t ctr) { return ctr * MODCR_DROP_WEYL; }
__device__ __forceinline__ void attn_drop_words_cm(uint32_t cm, uint32_t s0, uint32_t s1, uint32_t& x, uint32_t& y) {
    x = cm ^ s0;
    x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
    y = x * 0xC2B2AE35u + s1;
    y ^= y >> 16;
}
__device__ __forceinline__ void attn_drop_words(uint32_t ctr, uint32_t s0, uint32_t s1, uint32_t& x, uint32_t& y) {
    attn_drop_words_cm(attn_drop_cm(ctr), s0, s1, x, y);
}
// 0xffff in each 16-bit lane of w whose 15-bit uniform is >= thr15 (packed 16-bit subtract + arithmetic shift;
// thrm1_2 = (thr15 - 1) * 0x00010001)
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t attn_keep2(uint32_t w, uint32_t thrm1_2) {
    const s16x2_t u = __builtin_bit_cast(s16x2_t, w & 0x7fff7fffu);
    s16x2_t d = __builtin_bit_cast(s16x2_t, thrm1_2) - u;           // < 0 iff u >= thr15
    d = d >> 15;
    return __builtin_bit_cast(uint32_t, d);
}
__device__ __forceinline__ void attn_drop4_cm(uint32_t& d0, uint32_t& d1, uint32_t cm, uint32_t s0, uint32_t s1, uint32_t thrm1_2) {
    uint32_t x, y;
    attn_drop_words_cm(cm, s0, s1, x, y);
    d0 &= attn_keep2(x, thrm1_2);
    d1 &= attn_keep2(y, thrm1_2);
}
__device__ __forceinline__ void attn_drop4(uint32_t& d0, uint32_t& d1, uint32_t ctr, uint32_t s0, uint32_t s1, uint32_t thrm1_2) {
    attn_drop4_cm(d0, d1, attn_drop_cm(ctr), s0, s1, thrm1_2);
}
// the keep decision of key (4 g + f), f = 0..3, from the words of group g
__device__ __forceinline__ bool attn_keep_field(uint32_t x, uint32_t y, int f, uint32_t thr15) {
    const uint32_t w = (f & 2) ? y : x;
    return ((w >> ((f & 1) * 16)) & 0x7fffu) >= thr15;
}
// value of lane (quad base + E) in every lane of the quad (DPP quad_perm broadcast)
template <int E> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, E * 0x55, 0xf, 0xf, true);
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// cm = attn_drop_cm(counter of the lane's first key group): the second group is four counters further
__device__ __forceinline__ bf16x8 attn_drop8_cm(bf16x8 pb, uint32_t cm, uint32_t s0, uint32_t s1, uint32_t thr2) {
    u32x4_t w = __builtin_bit_cast(u32x4_t, pb);
    uint32_t a = w[0], b = w[1], c = w[2], d = w[3];
    attn_drop4_cm(a, b, cm, s0, s1, thr2);                          // keys 16 kb + 4 l4 + 0..3 of kb = 0
    attn_drop4_cm(c, d, cm + 4u * MODCR_DROP_WEYL, s0, s1, thr2);   // the same lane group's keys of kb = 1 (16 keys = 4 groups further)
    w[0] = a; w[1] = b; w[2] = c; w[3] = d;
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ bf16x8 attn_drop8(bf16x8 pb, uint32_t ctr, uint32_t s0, uint32_t s1, uint32_t thr2) {
    return attn_drop8_cm(pb, attn_drop_cm(ctr), s0, s1, thr2);
}

__device__ __forceinline__ int swz128(int row, int chunk) { return (row << 7) + (((chunk ^ (row >> 1)) & 7) << 4); }
__device__ __forceinline__ int swz64(int row, int chunk) { return (row << 6) + (((chunk ^ (row >> 2)) & 3) << 4); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr float LOG2E = 1.44269504088896340736f;


}  // namespace

struct AttnBwdArgs {
    const float* qkv; const void* dctx; const float* key_mask; const uint32_t* bits; float* dqkv;
    const bf16* qkvb;               // MFMA kernel: the recomputed q | k | v rows as bf16 (what the forward's images hold), qkv unused
    int out_bf16;                   // MFMA kernel: dq | dk | dv rows leave as bf16 (the operand dtype of the GEMMs that consume them)
    // attention-probability dropout of the forward (MFMA kernel only): 0 = off, else round(p * 2^15); hash keys; 1 / (1 - p);
    // token tile of the forward kernel (128 or 192: part of its counter layout)
    uint32_t drop_thr15, drop_s0, drop_s1;
    float drop_keep;
    int drop_lp;
    // gradient of the head-summed text -> region map (align map of modcr_qkv_attn_fwd) [N, T, R], or NULL: added to dP of
    // every head for query < T, key >= T (the map sums the UNMASKED probabilities)
    const float* d_align;
    int align_t;
    // five-product core (attn_bwd.hip): the forward's context rows [N, S, H] (delta = rowsum(dO o O)) and its log2-domain
    // row statistics lse[N, A, S] = log2(sum_j exp2(score_ij log2e)) (modcr_qkv_attn_lse_fwd); NULL = the older cores recompute them
    const bf16* ctx;
    const float* lse;
    // Q | K | V images the forward dumped (modcr_qkv_attn_lse_fwd's qkv_dump: [N][A][3][LP][64], Q scaled and chunk-averaged), or NULL
    const bf16* dump;
    // with dump + d_align: delta_align[N, A, S] = sum_j P_ij d_align_ij (text query i, region keys j), written by attn_dalign_delta_kernel
    float* delta_align;
    int N, S, H, A;
    int debug;      // tuning build only (MODCR_ATTN_BWD_DEBUG): 1 = return once the first images are built, 2 = no sub-pass Q, 4 = no sub-pass K
};

// attn_bwd.hip: launches attn_bwd5_kernel on `b` (bf16 rows, 0 < S <= 192, lse and ctx given, no align-map gradient)
__attribute__((visibility("hidden"))) int modcr_launch_attn_bwd5(const AttnBwdArgs& b, hipStream_t stream);
