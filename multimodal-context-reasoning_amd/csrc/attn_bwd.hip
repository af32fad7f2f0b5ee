// Attention core backward, five MFMA products per (query block, key block) instead of the older core's eight
// (CDNA4 / gfx950; bf16 rows, head size 64, 0 < S <= 192).
//
// Autograd of CaptionBertSelfAttention.forward (modeling_bert.py:46-72; v10:79-104) from the recomputed q | k | v rows,
// the forward's context rows O and its row statistics lse (log2 domain, written by modcr_qkv_attn_lse_fwd):
//     P  = exp2(Qs.K^T + mask - lse)          Qs = q log2e / 8 rounded to bf16: the values the forward's Q image held
//     dP = m o (dO.V^T)                       m  = the forward's dropout mask x 1 / (1 - p), regenerated from its counters
//     dS = P o (dP - delta)                   delta_i = sum_j P_ij dP_ij = dO_i . O_i
//     dV = (P o m)^T.dO     dK = dS^T.Qs / log2e     dQ = dS.K / 8
//
// One workgroup = 4 waves = one (sequence, head); wave w owns keys [WK w, WK w + WK) (WK = 48 on the 192-token tile, 32
// on the 128-token tile) for the whole tile: dK^T and dV^T of its keys stay in its accumulators while the workgroup sweeps
// the sequence in 32-query blocks, so neither needs a sum across waves.  Per block:
//   * S and dP are computed with the KEY ON THE LANE (A = Qs / dO rows of the block from LDS, B = the wave's K rows from
//     the K image / V rows held in registers).  Row constants ride in as the accumulator init (mask - lse; -delta), so
//     P = exp2(acc) needs no row maximum and no subtraction.  The accumulators of the two 16-query tiles, converted pairwise
//     to bf16, ARE the B operand of dV^T += dO^T.P and dK^T += Qs^T.dS (accumulator as operand: the 32 queries arrive in the
//     order 4 g + e | 16 + 4 g + e, and the A fragments -- ds_read_b64_tr_b16 of the same row images -- use that order).
//   * only dS crosses LDS, once: 8-byte pieces into a [key][32 queries] image (two buffers), and after ONE barrier per
//     block every wave computes dQ^T of the block for its 16 features from the whole image and the K image
//     (both operands by transposed reads), and stores 8-byte row pieces.
// The next block's Q / dO / O pieces are in flight (registers) under the current block's arithmetic.
// Padding: Q, dO, K, V rows beyond S are zero, keys beyond S carry a -inf mask (P = 0), lse / delta of rows beyond S are 0.
#include "common.h"
#include "attn_common.h"

namespace {

template <int KT>
struct AB5 {
    static constexpr int LP = 64 * KT;                      // token tile: 192 (KT = 3) or 128 (KT = 2)
    static constexpr int WK = 16 * KT;                      // keys per wave
    static constexpr int NT = 256;
    static constexpr int NKS = LP / 32;                     // 32-key steps of the dQ product = mask words per query row
    static constexpr int K_IMG = LP * 128;                  // [key][64] bf16, 128-byte rows, swz128
    static constexpr int QB = 32 * 128;                     // one block image: [32 queries][64]
    static constexpr int DS = LP * 64;                      // [key][32 queries] bf16, 64-byte rows, 8-byte pieces swizzled
    static constexpr int OFF_Q = K_IMG;
    static constexpr int OFF_DO = OFF_Q + 2 * QB;
    static constexpr int OFF_DS = OFF_DO + 2 * QB;
    static constexpr int OFF_LSE = OFF_DS + 2 * DS;         // -lse of every query row
    static constexpr int OFF_DL = OFF_LSE + LP * 4;         // -delta of the block's rows, two buffers
    static constexpr int OFF_BITS = OFF_DL + 2 * 32 * 4;    // dense-mask words of the block, [word][32 queries], two buffers
    static constexpr int SMEM = OFF_BITS + 2 * NKS * 32 * 4;
};

// 8-byte piece p (four queries) of row `key` of the dS image: pieces XOR-ed with (key & 7) ^ ((key >> 3) & 1), which spreads
// the 16 keys x 8 bytes of one ds_write_b64 lane group over all 32 banks and the 8 rows x 32 bytes of a transposed read over 64
__device__ __forceinline__ int ds_off(int key, int p) { return (key << 6) + (((p ^ key ^ (key >> 3)) & 7) << 3); }

typedef __attribute__((address_space(3))) bf16x4* lds_tr4;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Transposed fragment of a 128-byte-row image: feature 16 db + l15, tokens tok0 + 4 g + {0..3} and tok0 + 16 + 4 g + {0..3}
// (the row order of the accumulator-as-operand products).  Within a 16-lane group lane i addresses token row (i >> 2), 8-byte
// piece (i & 3) of the 16-feature span and receives feature i.
__device__ __forceinline__ bf16x8 tr8(const unsigned char* img, int tok0, int db, int l15, int g) {
    const int r = tok0 + 4 * g + (l15 >> 2);
    const int ch = db * 2 + ((l15 & 3) >> 1), within = (l15 & 1) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr4)(img + swz128(r, ch) + within));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr4)(img + swz128(r + 16, ch) + within));
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
    return o;
}

// MASK: 0 = broadcast key mask, 1 = dense mask bits [N, S, ceil(S / 32)].  DROP: the forward's attention-probability dropout.
template <int KT, int MASK, int DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd5_kernel(AttnBwdArgs p) {
    typedef AB5<KT> T;
    constexpr int LP = T::LP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;
    float* sLse = reinterpret_cast<float*>(smem + T::OFF_LSE);
    float* sDl = reinterpret_cast<float*>(smem + T::OFF_DL);
    uint32_t* sBits = reinterpret_cast<uint32_t*>(smem + T::OFF_BITS);
    const int S = p.S, H = p.H;
    const int n = blockIdx.x / p.A, a = blockIdx.x % p.A;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15_ = lane & 15, g_ = lane >> 4;
    const int LW = (S + 31) >> 5;
    const int nblk = (S + 31) >> 5;                         // 32-query blocks that hold a row of the sequence
    const bf16* qkv = p.qkvb + (int64_t)n * S * 3 * H + a * 64;
    const bf16* dctx = reinterpret_cast<const bf16*>(p.dctx) + (int64_t)n * S * H + a * 64;
    const bf16* octx = p.ctx + (int64_t)n * S * H + a * 64;
    bf16* dqkv = reinterpret_cast<bf16*>(p.dqkv) + (int64_t)n * S * 3 * H + a * 64;
    constexpr float QS = 0.125f * LOG2E;
    const int wkey0 = wave * T::WK;

    auto zero8 = [] {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
        return o;
    };
    // 16-byte piece c of token t's q / k / v row (part 0 / 1 / 2), zeros beyond S
    auto row8 = [&](int t, int part, int c) {
        return t < S ? *reinterpret_cast<const bf16x8*>(qkv + (int64_t)t * 3 * H + part * H + c * 8) : zero8();
    };

    // ---- the tile's tables: K image, this wave's V fragments and key masks, -lse ---------------------------------
#pragma unroll
    for (int i = 0; i < 2 * KT; ++i) {
        const int item = tid + T::NT * i, t = item >> 3, c = item & 7;
        *reinterpret_cast<bf16x8*>(sK + swz128(t, c)) = row8(t, 1, c);
    }
    bf16x8 fv[KT][2];
    float mkey[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = wkey0 + kt * 16 + l15_;
        fv[kt][0] = row8(key, 2, g_);
        fv[kt][1] = row8(key, 2, 4 + g_);
        if (key >= S) mkey[kt] = -INFINITY;
        else if (MASK) mkey[kt] = 0.f;
        else mkey[kt] = (1.0f - p.key_mask[(int64_t)n * S + key]) * (MODCR_NEG * LOG2E);
    }
    if (tid < LP) sLse[tid] = tid < S ? -p.lse[((int64_t)n * p.A + a) * S + tid] : 0.f;

    // ---- block staging: one 16-byte piece of Q, dO and O per thread (row r, piece c), one mask word per thread ---
    // (every per-lane index below is taken from a per-iteration opaque copy of the thread id: the LDS / global addresses
    // built from them are loop invariant, and hoisted out of the block loop they were spilled -- 56 to 70 registers)
    struct Blk { bf16x8 q, d, o; uint32_t w; };
    auto load_block = [&](int it, int tid) {
        const int br = tid >> 3, bc = tid & 7;
        Blk b;
        const int row = it * 32 + br;
        if (row < S) {
            b.q = *reinterpret_cast<const bf16x8*>(qkv + (int64_t)row * 3 * H + bc * 8);
            b.d = *reinterpret_cast<const bf16x8*>(dctx + (int64_t)row * H + bc * 8);
            b.o = *reinterpret_cast<const bf16x8*>(octx + (int64_t)row * H + bc * 8);
        } else {
            b.q = zero8(); b.d = zero8(); b.o = zero8();
        }
        b.w = 0xffffffffu;
        if (MASK) {
            const int wi = tid >> 5, ql = it * 32 + (tid & 31);
            if (wi < LW && ql < S) b.w = p.bits[((int64_t)n * S + ql) * LW + wi];
        }
        return b;
    };
    auto stage_block = [&](int buf, const Blk& b, int tid) {
        const int br = tid >> 3, bc = tid & 7;
        bf16x8 qs;
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            qs[e] = (bf16)((float)b.q[e] * QS);
            dot = fmaf((float)b.d[e], (float)b.o[e], dot);
        }
        *reinterpret_cast<bf16x8*>(smem + T::OFF_Q + buf * T::QB + swz128(br, bc)) = qs;
        *reinterpret_cast<bf16x8*>(smem + T::OFF_DO + buf * T::QB + swz128(br, bc)) = b.d;
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        if (bc == 0) sDl[buf * 32 + br] = -dot;
        if (MASK && tid < T::NKS * 32) sBits[buf * T::NKS * 32 + tid] = b.w;
    };

    f32x4 dk[KT][4], dv[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int db = 0; db < 4; ++db) { dk[kt][db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kt][db] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    {
        const Blk b0 = load_block(0, tid);
        stage_block(0, b0, tid);
    }
    __syncthreads();

#pragma unroll 1
    for (int it = 0; it < nblk; ++it) {
        const int buf = it & 1;
        const bool more = it + 1 < nblk;
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int l15 = tq & 15, g = (tq & 63) >> 4;
        Blk nb;
        if (more) nb = load_block(it + 1, tq);              // in flight under this block's arithmetic

        const unsigned char* q_img = smem + T::OFF_Q + buf * T::QB;
        const unsigned char* do_img = smem + T::OFF_DO + buf * T::QB;
        unsigned char* ds_img = smem + T::OFF_DS + buf * T::DS;
        bf16x8 fq[2][2], fdo[2][2];
        f32x4 nl[2], nd[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fq[qt][ks] = *reinterpret_cast<const bf16x8*>(q_img + swz128(qt * 16 + l15, ks * 4 + g));
                fdo[qt][ks] = *reinterpret_cast<const bf16x8*>(do_img + swz128(qt * 16 + l15, ks * 4 + g));
            }
            nl[qt] = *reinterpret_cast<const f32x4*>(sLse + it * 32 + qt * 16 + 4 * g);
            nd[qt] = *reinterpret_cast<const f32x4*>(sDl + buf * 32 + qt * 16 + 4 * g);
        }
        bf16x8 pB[KT], dsB[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int krow = wkey0 + kt * 16;
            const bf16x8 fk0 = *reinterpret_cast<const bf16x8*>(sK + swz128(krow + l15, g));
            const bf16x8 fk1 = *reinterpret_cast<const bf16x8*>(sK + swz128(krow + l15, 4 + g));
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x4 c;
#pragma unroll
                for (int e = 0; e < 4; ++e) c[e] = mkey[kt] + nl[qt][e];
                if (MASK) {                                 // bit (key & 31) of word (key >> 5) of the four query rows
                    const u32x4 w4 = *reinterpret_cast<const u32x4*>(sBits + buf * T::NKS * 32 + (krow >> 5) * 32 + qt * 16 + 4 * g);
                    const int bit = (krow & 31) + l15;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!((w4[e] >> bit) & 1u)) c[e] += MODCR_NEG * LOG2E;
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[qt][0], fk0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[qt][1], fk1, c, 0, 0, 0);
                f32x4 dp = DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : nd[qt];
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdo[qt][0], fv[kt][0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdo[qt][1], fv[kt][1], dp, 0, 0, 0);
                bf16x4 ds4;
                if (DROP) {
                    // a lane's four values are four queries of ONE key = four hash counters; the four lanes of a quad (the keys
                    // of one key group) need the same four, so each hashes one and they are exchanged by quad broadcasts
                    uint32_t hx, hy;
                    const int qrow = it * 32 + qt * 16 + 4 * g + (l15 & 3), key = krow + l15;
                    attn_drop_words((uint32_t)((n * p.A + a) * 256 + qrow), key >> 2, p.drop_key, hx, hy);
                    const uint32_t xq[4] = {quad_bcast<0>(hx), quad_bcast<1>(hx), quad_bcast<2>(hx), quad_bcast<3>(hx)};
                    const uint32_t yq[4] = {quad_bcast<0>(hy), quad_bcast<1>(hy), quad_bcast<2>(hy), quad_bcast<3>(hy)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(c[e]);
                        const bool keep = attn_keep_field(xq[e], yq[e], l15 & 3, p.drop_thr16);
                        const float v = keep ? fmaf(dp[e], p.drop_keep, nd[qt][e]) : nd[qt][e];
                        pB[kt][4 * qt + e] = (bf16)(keep ? pe : 0.f);   // dV takes the masked probabilities (x 1 / (1 - p) at the end)
                        ds4[e] = (bf16)(pe * v);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(c[e]);
                        pB[kt][4 * qt + e] = (bf16)pe;
                        ds4[e] = (bf16)(pe * dp[e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dsB[kt][4 * qt + e] = ds4[e];
                *reinterpret_cast<bf16x4*>(ds_img + ds_off(krow + l15, qt * 4 + g)) = ds4;
            }
        }
        // dV^T += dO^T.P, dK^T += Qs^T.dS over the block's 32 queries (the transposed fragments are read once per block)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const bf16x8 qf = tr8(q_img, 0, db, l15, g), df = tr8(do_img, 0, db, l15, g);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                dv[kt][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, pB[kt], dv[kt][db], 0, 0, 0);
                dk[kt][db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, dsB[kt], dk[kt][db], 0, 0, 0);
            }
        }
        if (more) stage_block(buf ^ 1, nb, tq);
        __syncthreads();

        // ---- dQ^T of the block, features 16 wave .. 16 wave + 15: sum over all keys of K^T . dS^T ---------------------
        f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < T::NKS; ++ks) {
            {
                const bf16x8 ka = tr8(sK, ks * 32, wave, l15, g);
                const int r0 = ks * 32 + 4 * g + (l15 >> 2);
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const int pc = qt * 4 + (l15 & 3);
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr4)(ds_img + ds_off(r0, pc)));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr4)(ds_img + ds_off(r0 + 16, pc)));
                    bf16x8 dsf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { dsf[e] = lo[e]; dsf[4 + e] = hi[e]; }
                    dq[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, dsf, dq[qt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = it * 32 + qt * 16 + l15;
            if (q < S) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(dq[qt][e] * 0.125f);
                *reinterpret_cast<bf16x4*>(dqkv + (int64_t)q * 3 * H + wave * 16 + 4 * g) = o;
            }
        }
    }

    // ---- dK, dV rows of this wave's keys ------------------------------------------------------------------------------
    const float vscale = DROP ? p.drop_keep : 1.0f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int key = wkey0 + kt * 16 + l15_;
        if (key < S) {
            bf16* ob = dqkv + (int64_t)key * 3 * H + H + 4 * g_;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                bf16x4 ok, ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) { ok[e] = (bf16)(dk[kt][db][e] * (1.0f / LOG2E)); ov[e] = (bf16)(dv[kt][db][e] * vscale); }
                *reinterpret_cast<bf16x4*>(ob + db * 16) = ok;
                *reinterpret_cast<bf16x4*>(ob + H + db * 16) = ov;
            }
        }
    }
}

template <int KT, int MASK, int DROP>
int launch5(const AttnBwdArgs& b, hipStream_t st) {
    typedef AB5<KT> T;
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd5_kernel<KT, MASK, DROP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::SMEM);
        if (e != hipSuccess) {
            modcr_set_error("attn_bwd5: cannot reserve %d bytes of LDS: %s", T::SMEM, hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL((attn_bwd5_kernel<KT, MASK, DROP>), dim3(b.N * b.A), dim3(T::NT), T::SMEM, st, b);
    return modcr_check_launch("attn_bwd5");
}

// ---- align-map gradient, part 1: what it adds to delta ----------------------------------------------------------------------
// The align map of seq_enc's layers 9-11 (v10:982: head-summed text -> region probabilities) has a gradient d_align [N, T, R] of its
// own (v10:1067-1073); through the softmax it enters dP (text query, region key: + d_align, every head) and therefore
// delta_i = sum_j P_ij dP_ij gains  sum_{j >= T} P_ij d_align_ij  -- a sum over keys that no single compute wave of
// attn_bwd6_kernel sees.  This kernel forms it from the forward's dump: one workgroup of 4 waves per (sequence, head), a wave per
// 16-query tile, operands straight from the dump rows (no LDS), P = exp2(Qs.K^T + mask - lse).  ~6 GFLOP at the bench size.
template <int MASK>
__global__ __launch_bounds__(256) void attn_dalign_delta_kernel(AttnBwdArgs p, int LP) {
    const int S = p.S, A = p.A, T = p.align_t, R = S - T;
    const int tile = blockIdx.x, n = tile / A;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g4 = lane >> 4;
    const int LW = (S + 31) >> 5;
    const bf16* dq = p.dump + (int64_t)tile * 3 * LP * 64;
    const bf16* dkk = dq + LP * 64;
    float* out = p.delta_align + (int64_t)tile * S;
    for (int q = T + threadIdx.x; q < S; q += 256) out[q] = 0.f;        // region queries: no align-map term
    for (int qt = wave; qt * 16 < T; qt += 4) {
        const int qrow = min(qt * 16 + l15, S - 1);
        const bf16x8 fq0 = *reinterpret_cast<const bf16x8*>(dq + qrow * 64 + g4 * 8);
        const bf16x8 fq1 = *reinterpret_cast<const bf16x8*>(dq + qrow * 64 + (4 + g4) * 8);
        float nl[4], acc[4];
        uint32_t dbase[4];                                  // side_post_drop: dropout bases of this lane's four queries (its key's l4 group)
        const bool post = p.side_post_drop && p.drop_thr16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = qt * 16 + 4 * g4 + e;
            nl[e] = q < T ? -p.lse[(int64_t)tile * S + q] : -INFINITY;
            acc[e] = 0.f;
            dbase[e] = post ? attn_drop_base((uint32_t)(tile * 256 + q), (uint32_t)(l15 >> 2), p.drop_key) : 0u;
        }
        for (int kt = T >> 4; kt * 16 < S; ++kt) {
            const int key = kt * 16 + l15, kc = min(key, S - 1);
            const bf16x8 fk0 = *reinterpret_cast<const bf16x8*>(dkk + kc * 64 + g4 * 8);
            const bf16x8 fk1 = *reinterpret_cast<const bf16x8*>(dkk + kc * 64 + (4 + g4) * 8);
            const bool in = key >= T && key < S;
            const float mk = !in ? -INFINITY : (MASK ? 0.f : (1.0f - p.key_mask[(int64_t)n * S + key]) * (MODCR_NEG * LOG2E));
            f32x4 c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c[e] = mk + nl[e];
                if (MASK && in) {
                    const int q = min(qt * 16 + 4 * g4 + e, S - 1);
                    if (!((p.bits[((int64_t)n * S + q) * LW + (key >> 5)] >> (key & 31)) & 1u)) c[e] += MODCR_NEG * LOG2E;
                }
            }
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq0, fk0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq1, fk1, c, 0, 0, 0);
            const uint32_t dcw = post ? attn_drop_const(kt * 2 + ((l15 >> 1) & 1)) : 0u;      // word of key (16 kt + l15): j = 2 (key >> 4) + ((key >> 1) & 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = qt * 16 + 4 * g4 + e;
                float da = (in && q < T) ? p.d_align[((int64_t)n * T + q) * R + (key - T)] : 0.f;
                if (post) {                                 // the map summed P o m / (1 - p): the same mask on its gradient
                    const uint32_t w = attn_drop_fold(dbase[e], dcw, p.drop_key);
                    da = (int)(short)(w >> (16 * (l15 & 1))) >= (int)p.drop_thr16 - 32768 ? da * p.drop_keep : 0.f;
                }
                if (in && q < T) acc[e] = fmaf(__builtin_amdgcn_exp2f(c[e]), da, acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            const int q = qt * 16 + 4 * g4 + e;
            if (l15 == 0 && q < T) out[q] = v;
        }
    }
}

// =====================================================================================================================
// Second form (attn_bwd6_kernel): the same five products on the images the FORWARD dumped (modcr_qkv_attn_lse_fwd's qkv_dump:
// Q scaled and chunk-averaged, K, V as plain rows per (sequence, head)) -- no projection is recomputed -- with the work
// cut finer and the memory traffic on waves of its own:
//   * one workgroup per CU, persistent over (sequence, head) tiles; NC = LP / 16 COMPUTE waves, each the owner of ONE 16-key
//     tile (dK^T / dV^T of its keys: 32 accumulator registers), + 4 LOADER waves: 16 waves = 4 per SIMD at <= 128 registers.
//     (attn_bwd5_kernel: 4 waves x 3 key tiles at ~240 registers, 2 waves per SIMD; it measured latency-bound -- every
//     instruction of a block on one of two waves per SIMD -- and every added feature spilled.)
//   * the loader waves own every global load of the block stream and the dQ stores: Q / dO / O pieces of the block two ahead
//     of the arithmetic (plain 16-byte loads, three register sets in flight; delta = rowsum(dO o O) formed on the way into LDS),
//     the next tile's K image and lse row with the first block of a tile, the dQ rows of the block before (staged through LDS
//     by the compute waves, stored as whole 128-byte rows).  The compute waves load only their own V rows / key mask per tile
//     and store their dK / dV rows (through a wave-private LDS transposition, whole rows).
//   * ONE workgroup barrier per 32-query block: compute(g) | barrier | dQ(g) by compute waves 0..7 beside compute(g + 1).
//     Buffers: Q / dO / delta / mask words x 3 (block g computed, g + 1 staged, g + 2 being staged), dS x 2, dQ staging x 2,
//     K image and lse x 2 (tile parity).
template <int KT, int MASK = 1>
struct AB6 {
    static constexpr int LP = 64 * KT, NC = LP / 16, NT = (NC + 4) * 64, NKS = LP / 32;
    static constexpr int K_IMG = LP * 128, QB = 32 * 128, DS = LP * 64;
    static constexpr int OFF_Q = 2 * K_IMG;
    static constexpr int OFF_DO = OFF_Q + 3 * QB;
    static constexpr int OFF_DS = OFF_DO + 3 * QB;
    static constexpr int OFF_OUT = OFF_DS + 2 * DS;         // dQ^T of a block as [32 queries][64] bf16 rows (swz128), two buffers
    static constexpr int OFF_EP = OFF_OUT + 2 * QB;         // per compute wave: [16 keys][64] rows for the dK / dV transposition
    static constexpr int OFF_LSE = OFF_EP + NC * 2048;
    static constexpr int OFF_DL = OFF_LSE + 2 * LP * 4;
    static constexpr int OFF_BITS = OFF_DL + 3 * 32 * 4;
    static constexpr int SMEM = OFF_BITS + (MASK ? 3 * NKS * 32 * 4 : 0);   // 132 KB at KT = 3
};

// Row swizzle of attn_bwd6's Q / dO / K images ([row][64] bf16, 128-byte rows): 16-byte chunk c of row r sits at chunk c ^ (r & 7).
// The forward's swz128 key, (r >> 1) & 7, is free of bank conflicts for 16-byte ROW reads only: a transposed 8-byte read
// (ds_read_b64_tr_b16: 32 lanes = rows 4 g4 + (l15 >> 2) = 8 consecutive rows x the chunk pair {2 db, 2 db + 1} x two halves, one
// 256-byte bank row = two image rows) then finds rows r and r + 2 on the same 16-byte slots -- 2-way on every transposed read
// of the kernel (round-4 counters: SQ_LDS_BANK_CONFLICT = 28 % of SQ_LDS_IDX_ACTIVE).  With r & 7 the four rows of one parity
// among any eight take four different chunk pairs, and the row reads (lane groups {0-3, 12-15, 20-27}, ... of ds_read_b128) stay
// conflict-free: rows {0-3, 12-15} with chunk c and rows {4-11} with chunk c + 1 cover eight distinct slots per row parity.
__device__ __forceinline__ int swzT(int row, int chunk) { return (row << 7) + (((chunk ^ row) & 7) << 4); }

// 8-byte piece p (four queries) of row `key` of the dS image, second form: the XOR key is (key & 7) ^ ((key >> 3) & 1), i.e. a
// function of key % 16 -- adding a multiple of 16 to `key` moves the address by whole rows, so one offset per lane + constants
__device__ __forceinline__ int ds_off6(int key, int p) { return (key << 6) + (((p ^ key ^ ((key >> 3) & 1)) & 7) << 3); }

// two transposed 8-byte reads (rows r .. r + 3 and r + 16 .. r + 19 of a 16-column span) as one MFMA operand
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* lo_at, int hi_delta) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr4)(lo_at));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr4)(lo_at + hi_delta));
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
    return o;
}

// one workgroup barrier without the vmcnt(0) a __syncthreads() carries: the loader waves' prefetches and every wave's stores stay
// in flight across it (LDS traffic is what the barrier orders)
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DALIGN: the align map's gradient rides in (d_align added to dP on the text-query x region-key block, delta_align added to delta)
template <int KT, int MASK, int DROP, int DALIGN = 0>
__global__ __launch_bounds__((64 * KT / 16 + 4) * 64) void attn_bwd6_kernel(AttnBwdArgs p) {
    typedef AB6<KT, MASK> T;
    constexpr int LP = T::LP, NC = T::NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = p.S, H = p.H, A = p.A;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int LW = (S + 31) >> 5, nblk = LW;                // >= 3 (the launcher checks S > 64)
    const int ntiles = p.N * A, tstep = gridDim.x;
    const int64_t tile_elems = (int64_t)3 * LP * 64;       // one (sequence, head) of the dump

    if (wave >= NC) {
        // ================================ loader waves ================================================================
        const int lt = tid - NC * 64;
        const int br = lt >> 3, bc = lt & 7;
        const int aSt = swzT(br, bc), aSt128 = swz128(br, bc);    // Q / dO images; the dQ staging buffer keeps the forward's swizzle
        struct Blk { bf16x8 q, d, o; uint32_t w; float da; };
        struct KSet { bf16x8 k[2 * KT]; float l; };
        int i_tile = blockIdx.x, i_it = 0;                  // issue stream
        auto issue = [&](Blk& b, KSet& ks) {                // loads of the issue stream's next block (+ its tile's K image / lse at it = 0)
            b.w = 0xffffffffu;
            if (i_tile < ntiles && !(MODCR_DBG(p.debug & 1) && (i_tile != (int)blockIdx.x || i_it > 3))) {      // debug 1: timing only, no block loads after the first four
                const int n = i_tile / A, a = i_tile - n * A;
                const int row = min(i_it * 32 + br, S - 1);
                const bf16* dq = p.dump + (int64_t)i_tile * tile_elems;
                b.q = *reinterpret_cast<const bf16x8*>(dq + row * 64 + bc * 8);
                b.d = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.dctx) + ((int64_t)n * S + row) * H + a * 64 + bc * 8);
                b.o = *reinterpret_cast<const bf16x8*>(p.ctx + ((int64_t)n * S + row) * H + a * 64 + bc * 8);
                if (DALIGN) b.da = p.delta_align[(int64_t)i_tile * S + row];
                if (MASK) {
                    const int wi = lt >> 5, ql = min(i_it * 32 + (lt & 31), S - 1);
                    if (wi < LW) b.w = p.bits[((int64_t)n * S + ql) * LW + wi];
                }
                if (i_it == 0) {
#pragma unroll
                    for (int j = 0; j < 2 * KT; ++j) {      // K rows: LP x 8 pieces over 256 threads (rows beyond S: whatever the forward left)
                        const int item = lt + 256 * j;
                        ks.k[j] = *reinterpret_cast<const bf16x8*>(dq + LP * 64 + (item >> 3) * 64 + (item & 7) * 8);
                    }
                    ks.l = p.lse[(int64_t)i_tile * S + min(lt, S - 1)];        // (used at staging time: no arithmetic on it here, the loads stay in flight)
                }
            }
            if (++i_it == nblk) { i_it = 0; i_tile += tstep; }
        };
        int s_tile = blockIdx.x, s_it = 0, s_q3 = 0, s_kb = 0;   // staging stream: block, its buffer (mod 3), its tile's parity
        auto stage = [&](const Blk& b, const KSet& ks) {
            if (s_tile < ntiles) {
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) dot = fmaf((float)b.d[e], (float)b.o[e], dot);
                *reinterpret_cast<bf16x8*>(smem + T::OFF_Q + s_q3 * T::QB + aSt) = b.q;
                *reinterpret_cast<bf16x8*>(smem + T::OFF_DO + s_q3 * T::QB + aSt) = b.d;
                dot += __shfl_xor(dot, 1, 64);
                dot += __shfl_xor(dot, 2, 64);
                dot += __shfl_xor(dot, 4, 64);
                if (DALIGN) dot += b.da;
                if (bc == 0) reinterpret_cast<float*>(smem + T::OFF_DL)[s_q3 * 32 + br] = -dot;
                if (MASK && lt < T::NKS * 32) reinterpret_cast<uint32_t*>(smem + T::OFF_BITS)[s_q3 * T::NKS * 32 + lt] = b.w;
                if (s_it == 0) {
#pragma unroll
                    for (int j = 0; j < 2 * KT; ++j) {
                        const int item = lt + 256 * j;
                        *reinterpret_cast<bf16x8*>(smem + s_kb * T::K_IMG + swzT(item >> 3, item & 7)) = ks.k[j];
                    }
                    if (lt < LP) reinterpret_cast<float*>(smem + T::OFF_LSE)[s_kb * LP + lt] = lt < S ? -ks.l : -INFINITY;
                }
            }
            s_q3 = s_q3 == 2 ? 0 : s_q3 + 1;
            if (++s_it == nblk) { s_it = 0; s_tile += tstep; s_kb ^= 1; }
        };
        int q_tile = blockIdx.x, q_it = 0, q_par = 0;       // dQ rows of the block the compute waves finished one barrier ago
        auto store_dq = [&]() {
            const int row = q_it * 32 + br;
            if (row < S && !MODCR_DBG(p.debug & 32)) {
                const int n = q_tile / A, a = q_tile - n * A;
                const uint4 v = *reinterpret_cast<const uint4*>(smem + T::OFF_OUT + q_par * T::QB + aSt128);
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.dqkv) + ((int64_t)n * S + row) * 3 * H + a * 64 + bc * 8) = v;
            }
            q_par ^= 1;
            if (++q_it == nblk) { q_it = 0; q_tile += tstep; }
        };
        // number of blocks this workgroup computes
        int nb_total = 0;
        for (int t = blockIdx.x; t < ntiles; t += tstep) nb_total += nblk;
        Blk b0, b1, b2;
        KSet ks;
        // prologue: blocks 0 and 1 staged before the first barrier, block 2 and 3 in flight
        issue(b0, ks); issue(b1, ks);
        stage(b0, ks); stage(b1, ks);
        issue(b0, ks); issue(b1, ks);
        wg_barrier();                                       // B(-1): blocks 0, 1 and the first tile's K image / lse are in LDS
        // interval g = [B(g - 1), B(g)]: compute(g) runs; here: dQ rows of block g - 2 leave, block g + 2 is staged, block g + 4 issued
        // (three register sets rotate: b0 = block g + 2, b1 = g + 3, b2 = g + 4)
        // (the store last: the staging's wait for its loads would otherwise also wait for a store issued just before it)
        for (int g = 0; g < nb_total; g += 3) {
            stage(b0, ks); issue(b2, ks);
            if (g >= 2) store_dq();
            wg_barrier();
            if (g + 1 >= nb_total) break;
            stage(b1, ks); issue(b0, ks);
            if (g + 1 >= 2) store_dq();
            wg_barrier();
            if (g + 2 >= nb_total) break;
            stage(b2, ks); issue(b1, ks);
            store_dq();
            wg_barrier();
        }
        // after the last barrier B(nb_total - 1): dQ(nb_total - 1) is being written by the compute waves; one more barrier publishes it
        if (nb_total >= 2) store_dq();                      // block nb_total - 2
        wg_barrier();                                       // B(nb_total)
        store_dq();                                         // block nb_total - 1
        return;
    }

    // ==================================== compute waves ================================================================
    const int key0 = wave * 16;                             // this wave's key tile
    const int aRow0 = swzT(l15, g4), aRow1 = swzT(l15, 4 + g4);
    int aTr[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) aTr[db] = swzT(4 * g4 + (l15 >> 2), db * 2 + ((l15 & 3) >> 1)) + (l15 & 1) * 8;
    const int dt = wave & 3, qtw = (wave >> 2) & 1;         // dQ phase (waves 0..7): feature block, 16-query tile
    const int aTrW = swzT(4 * g4 + (l15 >> 2), dt * 2 + ((l15 & 3) >> 1)) + (l15 & 1) * 8;
    const int aDsW0 = ds_off6(l15, g4) + key0 * 64, aDsW1 = ds_off6(l15, 4 + g4) + key0 * 64;
    const int aDsR = ds_off6(4 * g4 + (l15 >> 2), qtw * 4 + (l15 & 3));
    const int aOut = swz128(qtw * 16 + l15, (16 * dt + 4 * g4) >> 3) + ((16 * dt + 4 * g4) & 7) * 2;
    unsigned char* sEp = smem + T::OFF_EP + wave * 2048;

    bf16x8 fv0, fv1, fk0, fk1;
    float mkey;
    // (the loaded mask value is used only at the swap: arithmetic on it here would make the wave wait for the load)
    auto load_v = [&](int tile, bf16x8& v0, bf16x8& v1, float& mraw) {
        const int n = tile / A;
        const int kc = min(key0 + l15, S - 1);
        const bf16* dv_ = p.dump + (int64_t)tile * tile_elems + 2 * LP * 64;
        v0 = *reinterpret_cast<const bf16x8*>(dv_ + kc * 64 + g4 * 8);
        v1 = *reinterpret_cast<const bf16x8*>(dv_ + kc * 64 + (4 + g4) * 8);
        mraw = MASK ? 1.0f : p.key_mask[(int64_t)n * S + kc];
    };
    auto key_mask_of = [&](float mraw) { return key0 + l15 < S ? (1.0f - mraw) * (MODCR_NEG * LOG2E) : -INFINITY; };
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    load_v(blockIdx.x, fv0, fv1, mkey);
    mkey = key_mask_of(mkey);
    wg_barrier();                                           // B(-1)

    int q3 = 0, par = 0, kb = 0;
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += tstep, kb ^= 1) {
        const int n = tile / A, a = tile - n * A;
        const unsigned char* sK = smem + kb * T::K_IMG;
        const float* sLse = reinterpret_cast<const float*>(smem + T::OFF_LSE) + kb * LP + 4 * g4;
        fk0 = *reinterpret_cast<const bf16x8*>(sK + aRow0 + key0 * 128);
        fk1 = *reinterpret_cast<const bf16x8*>(sK + aRow1 + key0 * 128);
        // dropout (attn_common.h): Weyl product of the base counter of (query 4 g4 + (l15 & 3), this lane's l4 group) + per-block
        // strides below; the two word multipliers of this wave's 16-key block
        const uint32_t cm0 = (uint32_t)((((n * A + a) * 256 + 4 * g4 + (l15 & 3)) * 4) + (((key0 + l15) >> 2) & 3)) * MODCR_DROP_WEYL;
        const uint32_t cm16 = (uint32_t)(16 * 4) * MODCR_DROP_WEYL;                     // 16 queries further
        const uint32_t dcx = attn_drop_const((key0 >> 4) * 2), dcy = attn_drop_const((key0 >> 4) * 2 + 1);
        bf16x8 nv0, nv1;                                    // the next tile's V rows / key mask, loaded under the last block
        float nmk = 0.f;
#pragma unroll 1
        for (int it = 0; it < nblk; ++it) {
            const unsigned char* q_img = smem + T::OFF_Q + q3 * T::QB;
            const unsigned char* do_img = smem + T::OFF_DO + q3 * T::QB;
            unsigned char* ds_img = smem + T::OFF_DS + par * T::DS;
            const float* sDl = reinterpret_cast<const float*>(smem + T::OFF_DL) + q3 * 32 + 4 * g4;
            if (it == nblk - 1 && tile + tstep < ntiles) load_v(tile + tstep, nv0, nv1, nmk);
            bf16x8 pB, dsB;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float dal[4] = {0.f, 0.f, 0.f, 0.f};        // d_align of this lane's (query, key) pairs: text query x region key, else 0
                if (DALIGN) {
                    const int Ta = p.align_t, key = key0 + l15;
                    if (key >= Ta && key < S) {
                        const float* da = p.d_align + ((int64_t)n * Ta + it * 32 + qt * 16 + 4 * g4) * (S - Ta) + (key - Ta);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (it * 32 + qt * 16 + 4 * g4 + e < Ta) dal[e] = da[e * (S - Ta)];
                    }
                }
                const bf16x8 fq0 = *reinterpret_cast<const bf16x8*>(q_img + aRow0 + qt * 2048);
                const bf16x8 fq1 = *reinterpret_cast<const bf16x8*>(q_img + aRow1 + qt * 2048);
                const bf16x8 fd0 = *reinterpret_cast<const bf16x8*>(do_img + aRow0 + qt * 2048);
                const bf16x8 fd1 = *reinterpret_cast<const bf16x8*>(do_img + aRow1 + qt * 2048);
                const f32x4 nl = *reinterpret_cast<const f32x4*>(sLse + it * 32 + qt * 16);
                const f32x4 nd = *reinterpret_cast<const f32x4*>(sDl + qt * 16);
                f32x4 c;
#pragma unroll
                for (int e = 0; e < 4; ++e) c[e] = mkey + nl[e];
                if (MASK) {                                 // bit (key & 31) of word (key >> 5) of the four query rows
                    const u32x4 w4 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint32_t*>(smem + T::OFF_BITS) + q3 * T::NKS * 32 +
                                                                     (key0 >> 5) * 32 + qt * 16 + 4 * g4);
                    const int bit = (key0 & 31) + l15;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!((w4[e] >> bit) & 1u)) c[e] += MODCR_NEG * LOG2E;
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq0, fk0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq1, fk1, c, 0, 0, 0);
                f32x4 dp = DROP ? f32x4{0.f, 0.f, 0.f, 0.f} : nd;
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd0, fv0, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd1, fv1, dp, 0, 0, 0);
                bf16x4 ds4;
                if (MODCR_DBG(p.debug & 2)) {               // timing only: no exponentials, no mask
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pB[4 * qt + e] = (bf16)c[e]; ds4[e] = (bf16)dp[e]; }
                } else if (DROP) {
                    // a lane's four values are four queries of ONE key = four hash counters; the four lanes of a quad (the keys of
                    // one key group) need the same four, so each hashes one and they are exchanged by quad broadcasts
                    uint32_t hx, hy;
                    const uint32_t dbase = attn_drop_fold(cm0 + (uint32_t)(2 * it + qt) * cm16, 0x85EBCA6Bu, p.drop_key);
                    hx = attn_drop_fold(dbase, dcx, p.drop_key);
                    hy = attn_drop_fold(dbase, dcy, p.drop_key);
                    const uint32_t xq[4] = {quad_bcast<0>(hx), quad_bcast<1>(hx), quad_bcast<2>(hx), quad_bcast<3>(hx)};
                    const uint32_t yq[4] = {quad_bcast<0>(hy), quad_bcast<1>(hy), quad_bcast<2>(hy), quad_bcast<3>(hy)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(c[e]);
                        const bool keep = attn_keep_field(xq[e], yq[e], l15 & 3, p.drop_thr16);
                        // the map summed the un-dropped probabilities, or (side_post_drop) the dropped ones: d_align under the mask
                        float v = keep ? fmaf((DALIGN && p.side_post_drop) ? dp[e] + dal[e] : dp[e], p.drop_keep, nd[e]) : nd[e];
                        if (DALIGN && !p.side_post_drop) v += dal[e];
                        pB[4 * qt + e] = (bf16)(keep ? pe : 0.f);       // dV takes the masked probabilities (x 1 / (1 - p) at the end)
                        ds4[e] = (bf16)(pe * v);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(c[e]);
                        pB[4 * qt + e] = (bf16)pe;
                        ds4[e] = (bf16)(pe * (DALIGN ? dp[e] + dal[e] : dp[e]));
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dsB[4 * qt + e] = ds4[e];
                *reinterpret_cast<bf16x4*>(ds_img + (qt ? aDsW1 : aDsW0)) = ds4;
            }
            if (!MODCR_DBG(p.debug & 8))
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8 qf = tr_pair(q_img + aTr[db], 2048), df = tr_pair(do_img + aTr[db], 2048);
                dv[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, pB, dv[db], 0, 0, 0);
                dk[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, dsB, dk[db], 0, 0, 0);
            }
            if (it == nblk - 1 && tile + tstep < ntiles)      // the next tile's V rows have landed (the only loads in flight: no store yet)
                asm volatile("" : "+v"(nv0), "+v"(nv1), "+v"(nmk));
            wg_barrier();                                   // B(g): dS of this block complete

            // ---- dQ^T of the block, one 16 x 16 tile per wave 0..7: features 16 dt.., queries 16 qtw.. -------------------------
            if (wave < 8 && !MODCR_DBG(p.debug & 4)) {
                f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < T::NKS; ++ks) {
                    const bf16x8 ka = tr_pair(sK + aTrW + ks * 32 * 128, 2048);
                    const bf16x8 dsf = tr_pair(ds_img + aDsR + ks * 32 * 64, 1024);
                    dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, dsf, dq, 0, 0, 0);
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(dq[e] * 0.125f);
                *reinterpret_cast<bf16x4*>(smem + T::OFF_OUT + par * T::QB + aOut) = o;
            }
            if (it == nblk - 1) {               // (behind the barrier and the dQ phase: nothing waits for these stores)
                // ---- dK, dV rows of this wave's keys: through the wave's own LDS rows, out as whole 128-byte rows -----------
                const float vscale = DROP ? p.drop_keep : 1.0f;
                bf16* ob = reinterpret_cast<bf16*>(p.dqkv) + ((int64_t)n * S + key0) * 3 * H + a * 64;
#pragma unroll
                for (int part = 1; part <= 2; ++part) {
#pragma unroll
                    for (int db = 0; db < 4; ++db) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = part == 1 ? (bf16)(dk[db][e] * (1.0f / LOG2E)) : (bf16)(dv[db][e] * vscale);
                        const int d0 = 16 * db + 4 * g4;
                        *reinterpret_cast<bf16x4*>(sEp + swz128(l15, d0 >> 3) + (d0 & 7) * 2) = o;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int r = hf * 8 + (lane >> 3), cc = lane & 7;
                        const uint4 v = *reinterpret_cast<const uint4*>(sEp + swz128(r, cc));
                        if (key0 + r < S && !MODCR_DBG(p.debug & 128))
                            *reinterpret_cast<uint4*>(ob + (int64_t)r * 3 * H + part * H + cc * 8) = v;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
#pragma unroll
                for (int db = 0; db < 4; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                if (tile + tstep < ntiles) { fv0 = nv0; fv1 = nv1; mkey = key_mask_of(nmk); }
            }
            q3 = q3 == 2 ? 0 : q3 + 1;
            par ^= 1;
        }
    }
    wg_barrier();                                           // B(nb_total): the last block's dQ rows are in LDS for the loader waves
}

template <int KT, int MASK, int DROP, int DALIGN = 0>
int launch6(const AttnBwdArgs& b, hipStream_t st) {
    typedef AB6<KT, MASK> T;
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd6_kernel<KT, MASK, DROP, DALIGN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::SMEM);
        if (e != hipSuccess) {
            modcr_set_error("attn_bwd6: cannot reserve %d bytes of LDS: %s", T::SMEM, hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    const int ncu = modcr_device_cus();
    const int ntiles = b.N * b.A;
    int grid = ntiles < ncu ? ntiles : ncu;                 // one resident workgroup per CU walks the tiles
    if (modcr_knob_set("MODCR_ATTN_BWD_GRID")) grid = modcr_knob_int("MODCR_ATTN_BWD_GRID", grid);      // tuning build only
    if (DALIGN) {                                           // part 1: the align map's share of delta
        if (MASK) hipLaunchKernelGGL((attn_dalign_delta_kernel<1>), dim3(ntiles), dim3(256), 0, st, b, (int)T::LP);
        else hipLaunchKernelGGL((attn_dalign_delta_kernel<0>), dim3(ntiles), dim3(256), 0, st, b, (int)T::LP);
        const int rc = modcr_check_launch("attn_dalign_delta");
        if (rc != MODCR_OK) return rc;
    }
    hipLaunchKernelGGL((attn_bwd6_kernel<KT, MASK, DROP, DALIGN>), dim3(grid), dim3(T::NT), T::SMEM, st, b);
    return modcr_check_launch("attn_bwd6");
}

template <int KT>
int launch6_kt(const AttnBwdArgs& b, hipStream_t st) {
    const bool drop = b.drop_thr16 != 0;
    if (b.d_align) {                                        // seq_enc layers 9-11: dense mask (the key-mask form exists for completeness)
        if (b.bits) return drop ? launch6<KT, 1, 1, 1>(b, st) : launch6<KT, 1, 0, 1>(b, st);
        return drop ? launch6<KT, 0, 1, 1>(b, st) : launch6<KT, 0, 0, 1>(b, st);
    }
    if (b.bits) return drop ? launch6<KT, 1, 1>(b, st) : launch6<KT, 1, 0>(b, st);
    return drop ? launch6<KT, 0, 1>(b, st) : launch6<KT, 0, 0>(b, st);
}

template <int KT>
int launch5_kt(const AttnBwdArgs& b, hipStream_t st) {
    const bool drop = b.drop_thr16 != 0;
    if (b.bits) return drop ? launch5<KT, 1, 1>(b, st) : launch5<KT, 1, 0>(b, st);
    return drop ? launch5<KT, 0, 1>(b, st) : launch5<KT, 0, 0>(b, st);
}

}  // namespace

int modcr_launch_attn_bwd5(const AttnBwdArgs& b, hipStream_t stream) {
    MODCR_REQUIRE((b.qkvb || b.dump) && b.dctx && b.ctx && b.lse && b.dqkv && (b.key_mask || b.bits), "attn_bwd5: null pointer");
    MODCR_REQUIRE(b.S > 0 && b.S <= 192 && b.H == b.A * 64 && (!b.d_align || (b.dump && b.delta_align && b.S > 64)), "attn_bwd5: unsupported call (S=%d)", b.S);
    MODCR_REQUIRE(modcr_aligned16(b.qkvb) && modcr_aligned16(b.dctx) && modcr_aligned16(b.ctx) && modcr_aligned16(b.dqkv), "attn_bwd5: 16-byte alignment");
    if (b.dump && b.S > 64 && !modcr_knob_set("MODCR_ATTN_BWD_V5")) {            // the forward dumped its Q | K | V images
        MODCR_REQUIRE(modcr_aligned16(b.dump), "attn_bwd6: 16-byte alignment");
        return b.S <= 128 ? launch6_kt<2>(b, stream) : launch6_kt<3>(b, stream);
    }
    return b.S <= 128 ? launch5_kt<2>(b, stream) : launch5_kt<3>(b, stream);
}
