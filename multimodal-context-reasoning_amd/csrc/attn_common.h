// Helpers shared by the attention kernels (attn.hip: forward tile kernels + the older backward cores; attn_bwd.hip: the
// five-product backward core): the attention-probability dropout hash, LDS row swizzles, the backward argument block.
#pragma once
#include "common.h"

namespace {

// nn.Dropout on the softmax output (modeling_bert.py:69 / v10:101), counter-based, two levels (round 5):
//     base(row, l4)  = fold(cm * 0x85EBCA6B + K),  cm = (row * 4 + l4) * 0x9E3779B1 mod 2^32,  row = (n * A + head) * 256 + query
//     word(base, j)  = fold(base * C[j] + K),      fold(p) = lo32(p) ^ hi32(p) of the 64-bit sum (one v_mad_u64_u32 + one v_xor)
// K = the 64-bit key derived from (seed, offset) on the host.  Key `key` of a query row takes l4 = (key >> 2) & 3 (its group of
// four inside a 16-key block), word j = 2 * (key >> 4) + ((key >> 1) & 1) and the 16-bit field key & 1 of it (low / high half);
// the weight is kept iff that field, read as a signed 16-bit number, is >= thr16 - 32768 with thr16 = round(p * 2^16) -- one
// saturating packed subtract + arithmetic shift per word gives the 0xffff / 0 keep masks of a packed bf16 pair.
// C[j] = fmix32((j + 1) * 0x9E3779B1) | 1: a compile-time literal wherever j is (the unrolled key tiles of every MFMA kernel).
// The layout does not depend on the token tile, so every kernel (128 / 192 / 256-token tiles, the older kernel, the three backward
// cores) regenerates the same mask for P + S <= 256.
// Why this form: the round-3/4 hash (Weyl product, two xorshift-multiply rounds per group of four keys, 15-bit fields) was 19 VALU
// instructions per four decisions incl. the compare -- 686 of the ~900 VALU instructions of the forward's phase B, 7 % of the kernel;
// here a lane's base is hashed once per (query, l4) and tile, and every further word costs two instructions: 10 per eight decisions
// + 12 for the compare / mask = 22 against 38.  Statistics (tools/hash_stats.py: keep fraction per field, all 496 field pairs of a
// row's first 16 words, sampled triples, neighbouring l4 / query / head / sequence, cross-seed, histograms; 4 M rows x 3 seeds):
// pair-frequency z-scores rms 1.01, max 3.1 -- noise; a single multiply-fold of the Weyl counter used directly (no second level)
// fails the neighbour tests by 30 sigma, which is why the base is never used as a word.
#define MODCR_DROP_WEYL 0x9E3779B1u
typedef short s16x2_t __attribute__((ext_vector_type(2)));
__host__ __device__ constexpr uint32_t attn_drop_const(int j) {
    uint32_t x = (uint32_t)(j + 1) * 0x9E3779B1u;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x | 1u;
}
__device__ __forceinline__ uint32_t attn_drop_fold(uint32_t a, uint32_t c, uint64_t k) {
    const uint64_t p = (uint64_t)a * c + k;
    return (uint32_t)p ^ (uint32_t)(p >> 32);
}
// row = (n * A + head) * 256 + query (tile row: prefix rows count), l4 = 0..3
__device__ __forceinline__ uint32_t attn_drop_base(uint32_t row, uint32_t l4, uint64_t k) {
    return attn_drop_fold((row * 4u + l4) * MODCR_DROP_WEYL, 0x85EBCA6Bu, k);
}
__device__ __forceinline__ uint32_t attn_drop_word(uint32_t base, int j, uint64_t k) { return attn_drop_fold(base, attn_drop_const(j), k); }
// 0xffff in each 16-bit half of w that is kept; thrm1_2 = ((thr16 - 32769) & 0xffff) * 0x00010001
__device__ __forceinline__ uint32_t attn_keep2(uint32_t w, uint32_t thrm1_2) {
    s16x2_t d = __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2_t, thrm1_2), __builtin_bit_cast(s16x2_t, w));   // < 0 iff field >= thr
    d = d >> 15;
    return __builtin_bit_cast(uint32_t, d);
}
__host__ __device__ inline uint32_t attn_thr16(float p) {
    uint32_t t = (uint32_t)((double)p * 65536.0 + 0.5);
    return t < 1u ? 1u : (t > 65535u ? 65535u : t);
}
__host__ __device__ inline uint32_t attn_thrm1_2(uint32_t thr16) { return ((thr16 - 32769u) & 0xffffu) * 0x00010001u; }
// the two words of the 4-key group `g4` = key >> 2 (0..63) of a row: x decides keys 4 g4, 4 g4 + 1, y keys 4 g4 + 2, 4 g4 + 3
__device__ __forceinline__ void attn_drop_words(uint32_t row, int g4, uint64_t k, uint32_t& x, uint32_t& y) {
    const uint32_t b = attn_drop_base(row, (uint32_t)(g4 & 3), k);
    x = attn_drop_word(b, (g4 >> 2) * 2, k);
    y = attn_drop_word(b, (g4 >> 2) * 2 + 1, k);
}
// the keep decision of key (4 g4 + f), f = 0..3, from the words of its group
__device__ __forceinline__ bool attn_keep_field(uint32_t x, uint32_t y, int f, uint32_t thr16) {
    const uint32_t w = (f & 2) ? y : x;
    return (int)(short)(w >> ((f & 1) * 16)) >= (int)thr16 - 32768;
}
// value of lane (quad base + E) in every lane of the quad (DPP quad_perm broadcast)
template <int E> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, E * 0x55, 0xf, 0xf, true);
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// the eight weights of a lane in one 32-key tile kt (keys 32 kt + 16 kb + 4 l4 + e: pb[4 kb + e]); base = attn_drop_base(row, l4)
// (the multipliers are made opaque SGPR values: left visible, the compiler chains the words as P[j+1] = P[j] + base * (C[j+1] - C[j]) --
// a serial v_mad_u64_u32 chain plus a v_sub correction of the high word whenever the difference wraps)
template <int KT> __device__ __forceinline__ bf16x8 attn_drop8(bf16x8 pb, uint32_t base, uint64_t k, uint32_t thrm1_2) {
    u32x4_t w = __builtin_bit_cast(u32x4_t, pb);
#pragma unroll
    for (int i = 0; i < 4; ++i) {                           // i = 2 kb + (e >> 1)
        uint32_t c = attn_drop_const(4 * KT + i);
        asm volatile("" : "+s"(c));
        w[i] &= attn_keep2(attn_drop_fold(base, c, k), thrm1_2);
    }
    return __builtin_bit_cast(bf16x8, w);
}
// the same with a run-time tile index (the exact pass)
__device__ __forceinline__ bf16x8 attn_drop8_rt(bf16x8 pb, uint32_t base, int kt, uint64_t k, uint32_t thrm1_2) {
    u32x4_t w = __builtin_bit_cast(u32x4_t, pb);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] &= attn_keep2(attn_drop_word(base, 4 * kt + i, k), thrm1_2);
    return __builtin_bit_cast(bf16x8, w);
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
    // attention-probability dropout of the forward: 0 = off, else thr16 = round(p * 2^16); the 64-bit key; 1 / (1 - p)
    uint32_t drop_thr16;
    uint64_t drop_key;
    float drop_keep;
    int side_post_drop;             // MODCR_ATTN_SIDE_POST_DROPOUT: d_align belongs to a map of the probabilities after the dropout
    // gradient of the head-summed text -> region map (align map of modcr_qkv_attn_fwd) [N, T, R], or NULL: added to dP of
    // every head for query < T, key >= T (un-dropped probabilities; with side_post_drop under the forward's dropout mask)
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
