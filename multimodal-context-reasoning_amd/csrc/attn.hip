// Fused QKV projection + masked softmax attention for one encoder layer (CDNA4 / gfx950).
//
// bf16 path -- one workgroup per (sequence n, head a), NW = ceil((P+S)/32) rounded to {2,4,6,8}
// waves, wave w owns tile rows / queries 32w..32w+31:
//   phase A  [L_pad x 192] = [prefix;x] . [Wq_a;Wk_a;Wv_a]^T   (K = H, 64-wide K-tiles staged in
//            LDS as swizzled 128-byte rows, v_mfma_f32_32x32x16_bf16).  Q and K are produced with
//            the weights as the MFMA A operand (features in registers, tokens on lanes) so four
//            consecutive features pack into one 8-byte LDS store of a row-major [token][64] image;
//            V is produced the other way round and lands transposed, [64][token], which is what
//            the P.V product wants.  Q, K, V never touch HBM.
//   chunk-mean query (v10:66-78) on the Q image in LDS when chunk_id is given.
//   phase B  S^T = K.Q^T per 32-key tile (keys in registers, query on the lane): the softmax row
//            reduce is a register reduce + one cross-half exchange; the probabilities, converted
//            pairwise to bf16, ARE the B operand of O^T = V^T.P^T (no LDS round trip).
//   epilogue O^T * 1/rowsum -> bf16 -> wave-private LDS transpose -> 128-byte row stores.
// Masks follow the reference: additive -10000 for masked keys inside the sequence
// (modeling_transfomres.py:641), padding lanes beyond P+S are excluded outright.
//
// fp32 path (parity): QKV by the fp32 GEMM into a workspace, chunk-mean kernel, then a plain
// VALU attention core with identical mask semantics.
#include <stdlib.h>

#include <type_traits>

#include <hip/hip_ext.h>

#include "common.h"
#include "attn_common.h"

namespace {

// Measurement hook (modcr_time_next_attn): two hipEvent_t the NEXT fused-attention launch of this thread hands to
// hipExtLaunchKernel, which stamps them at the start and the end of the kernel itself -- the launch's duration as rocprofv3
// reports it.  (A hipEventRecord pair around the call adds the dispatch latency of two extra barrier packets: 5 % on a
// 370 us launch inside a training step.)
thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;
template <typename K, typename A>
inline void launch_timed(K kernel, dim3 grid, dim3 block, size_t smem, hipStream_t st, const A& args) {
    if (g_time_start && g_time_stop) {
        hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)smem, st, g_time_start, g_time_stop, 0, args);
        g_time_start = g_time_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, smem, st, args);
    }
}

struct AttnArgs {
    const bf16* x; const bf16* hist; const bf16* wqkv; const float* bqkv;
    const float* key_mask; const uint32_t* bits; const int32_t* chunk_id;
    bf16* ctx; float* probs; float* align_map;
    float* lse;     // tile kernels: log2-domain row statistics [N, A, S] for the five-product backward (attn_bwd.hip), or NULL
    bf16* dump;     // tile kernels (two heads per workgroup, no prefix rows): the Q (scaled, chunk-mean applied) | K | V images of
                    // every (sequence, head) as plain rows [N][A][3][LP][64] for the backward (instead of recomputing them), or NULL
    int N, S, P, H, A, chunk_t, align_t;
    int hconc;      // head groups an XCD works on at a time (block -> tile order), 0 = sequence-major
    unsigned long long* trace = nullptr;   // tuning build only (MODCR_ATTN_TRACE_PTR): cycle stamps of workgroup 0's waves 0 and 4 per tile
    int debug;      // tuning build only (MODCR_ATTN_DEBUG, compiled out of the product library): 1 = stop after phase A, 2 = skip the phase-A MFMA loop, 8 = force the exact pass
    // attention-probability dropout (training mode, attn_common.h): on / off; the packed threshold attn_thrm1_2(round(p * 2^16)),
    // the 64-bit key derived from (seed, offset), and 1 / (1 - p)
    int drop_on;
    uint32_t drop_thr2;
    uint64_t drop_key;
    float drop_keep;
    // side outputs (probabilities, align map) AFTER the dropout, P o m / (1 - p), as the reference returns them (modeling_bert.py:69-74,
    // v10:94-106: `attention_probs = self.dropout(attention_probs)` comes before the return) -- MODCR_ATTN_SIDE_POST_DROPOUT; 0 = the
    // un-dropped probabilities (same expectation)
    int side_post_drop;
    // LayerNorm fold PROTOTYPE (tuning library only): x holds the PRE-LayerNorm rows of the previous sublayer,
    // wqkv = bf16(gamma o W), bqkv = d = W beta + b, cfold[3H] = row sums of the folded weights, row_stats[N * L][2] = (rstd, rstd * mu)
    // of every row: the image pass forms rstd * acc - (rstd * mu) * c + d instead of acc + b
#ifdef MODCR_TUNING
    const float* row_stats = nullptr;
    const float* cfold = nullptr;
#endif
};

constexpr int VT_PAD = 8;  // bytes of padding per V^T row: stride/4 = 2*odd -> conflict-free b64 reads

// HPW = heads per workgroup.  A 6-wave workgroup (S <= 192) lands 2,2,1,1 on the four SIMDs and a
// second one only fits beside it if the dispatcher happens to rotate its start SIMD -- measured
// residency was ~1 workgroup per CU.  Two heads of the same sequence per workgroup give 12 waves
// = 3 per SIMD by construction, share the X tile of the QKV GEMM (half the L2->LDS traffic for X)
// and let one wave's softmax VALU run under its SIMD neighbours' MFMAs.
// OCC = minimum waves per SIMD the register allocator must leave room for.
// BKA / NSLOT: K-tile width (64 = full 128-byte lines, 32) and ring depth of the phase-A staging.
template <int NW, int HPW, int OCC, int BKA, int NSLOT>
__global__ __launch_bounds__(NW * HPW * 64, OCC) void qkv_attn_bf16_kernel(AttnArgs p) {
    constexpr int LP = NW * 32;              // padded key / tile-row count
    constexpr int NTH = NW * 64;             // threads per head
    constexpr int NT = NTH * HPW;            // threads
    constexpr int VT_STRIDE = LP * 2 + VT_PAD;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // phase A view: two ring slots of {X tile: LP rows x 128 B, W tile per head: 192 rows x 128 B}
    // phase B view (aliases phase A after a barrier)
    constexpr int ROWB = BKA * 2;                // bytes per staged row
    constexpr int STAGE = (LP + 192 * HPW) * ROWB; // bytes per ring slot: X tile + one W tile per head
    constexpr int END_A = NSLOT * STAGE, HEAD_B = 2 * LP * 128 + 64 * VT_STRIDE, END_B = HPW * HEAD_B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hd = wave_all / NW, wave = wave_all % NW;   // head within the workgroup, wave within the head
    const int ltid = tid - hd * NTH;
    unsigned char* sQ = smem + hd * HEAD_B;      // LP x 128 B, row = query index
    unsigned char* sK = sQ + LP * 128;           // LP x 128 B, row = key index
    unsigned char* sVt = sQ + 2 * LP * 128;      // 64 rows x VT_STRIDE
    float* sMask = reinterpret_cast<float*>(smem + (END_A > END_B ? END_A : END_B));  // LP floats, outside both views
    float* sBiasAll = sMask + LP;                                                     // 192 floats per head
    int* sCid = reinterpret_cast<int*>(sBiasAll + 192 * HPW);                         // LP ints
    float* sBias = sBiasAll + hd * 192;

    const int hgroups = p.A / HPW;
    const int nwg = p.N * hgroups;
    // Block -> (sequence, head group).  Blocks go round-robin over the 8 XCDs; each XCD owns a
    // contiguous run of sequences.  Inside an XCD the order is head-group-major over `hconc` groups at
    // a time, so the ~32 workgroups resident on the XCD share hconc weight slices (hconc x 576 KiB,
    // L2-resident) instead of all of Wqkv (3.4 MiB of the 4 MiB L2, evicted by the X stream).
    int n, hg;
    if (p.hconc > 0 && (p.N & 7) == 0) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, ns = p.N >> 3, G = p.hconc;
        const int hgb = idx / (ns * G), rem = idx - hgb * ns * G;
        const int gw = min(G, hgroups - hgb * G);          // last block of groups may be narrower
        n = xcd * ns + rem / gw;
        hg = hgb * G + rem % gw;
    } else {
        const int tile = xcd_remap(blockIdx.x, nwg);
        n = tile / hgroups, hg = tile % hgroups;
    }
    const int a0 = hg * HPW, a = a0 + hd;
    const int r = lane & 31, h = lane >> 5;
    const int S = p.S, P = p.P, L = P + S, H = p.H;

    // ---- small per-block tables ------------------------------------------------------------
    for (int j = tid; j < LP; j += NT) {
        // scores are kept in the log2 domain (Q is scaled by log2(e)/8), so the additive mask is too
        float m;
        if (j >= L) m = -INFINITY;
        else if (p.bits) m = 0.f;
        else m = (1.0f - p.key_mask[(int64_t)n * L + j]) * (MODCR_NEG * LOG2E);
        sMask[j] = m;
        sCid[j] = (p.chunk_id && j < p.chunk_t) ? p.chunk_id[(int64_t)n * p.chunk_t + j] : -1;
    }
    for (int j = tid; j < 192 * HPW; j += NT) {
        const int jh = j / 192, jj = j % 192;
        sBiasAll[j] = p.bqkv[(jj >> 6) * H + (a0 + jh) * 64 + (jj & 63)];
    }

    // ---- phase A: QKV tile GEMM ---------------------------------------------------------------
    // [LP tokens] x [192 features per head] over K = H.  K-tiles of 64 (128-byte rows) arrive by
    // LDS-DMA (8 rows = 1 KiB per instruction) into a two-slot ring: the DMA of tile k+1 is issued
    // right after the barrier that opens tile k and lands while tile k feeds the MFMAs.  LDS is
    // written linearly, so the XOR swizzle (slot s of row r holds logical chunk s ^ ((r>>1)&7))
    // is applied to the SOURCE address.  Within a head, wave w owns token blocks {2(w/2), 2(w/2)+1}
    // x feature blocks {3(w%2) .. 3(w%2)+2}: 5 fragment reads per 6 MFMAs.
    constexpr int NWAVES = NW * HPW;
    constexpr int RPC = 1024 / ROWB;            // rows per 1-KiB DMA piece (8 or 16)
    constexpr int CPRW = ROWB / 16;             // 16-byte chunks per row (8 or 4)
    constexpr int XCH = LP / RPC;               // pieces of the X tile
    constexpr int WCHH = 192 / RPC;             // pieces of one head's W tile
    constexpr int NCH = XCH + WCHH * HPW;
    constexpr int CPW = NCH / NWAVES;           // pieces (DMA instructions) per wave per K-tile
    static_assert(NCH % NWAVES == 0, "pieces must divide evenly over the waves");
    static_assert(STAGE == NCH * 1024, "stage size");
    static_assert((NSLOT - 1) * CPW < 64, "vmcnt field");
    auto skey = [](int row) { return BKA == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };
    auto soff = [](int row, int chunk) {
        return BKA == 64 ? (row << 7) + (((chunk ^ (row >> 1)) & 7) << 4) : (row << 6) + (((chunk ^ (row >> 2)) & 3) << 4);
    };
    const bf16* gsrc[CPW];
#pragma unroll
    for (int q = 0; q < CPW; ++q) {
        const int id = wave_all + q * NWAVES;
        if (id < XCH) {
            const int trow = id * RPC + lane / CPRW;
            const int row = min(trow, L - 1);                                // padding rows re-read row L-1
            const int c = (lane % CPRW) ^ skey(trow);
            const bf16* base = (row < P) ? p.hist + ((int64_t)n * P + row) * H
                                         : p.x + ((int64_t)n * S + (row - P)) * H;
            gsrc[q] = base + c * 8;
        } else {
            const int w = id - XCH, wh = w / WCHH;
            const int f = (w % WCHH) * RPC + lane / CPRW;                    // 0..191 = q|k|v feature of head a0+wh
            const int c = (lane % CPRW) ^ skey(f);
            gsrc[q] = p.wqkv + ((int64_t)(f >> 6) * H + (a0 + wh) * 64 + (f & 63)) * H + c * 8;
        }
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int q = 0; q < CPW; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[q] + k0),
                                             (lptr_t)(smem + buf * STAGE + (wave_all + q * NWAVES) * 1024), 16, 0, 0);
    };

    const int tg = wave >> 1, fg = wave & 1;
    f32x16 acc[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;

    // Ring of NSLOT K-tiles with NSLOT-1 in flight: top of iteration kt waits (counted vmcnt) until
    // tile kt has landed, raw barrier, refill of the slot everyone finished reading one iteration
    // ago, compute.  LDS-DMA pieces land ~1-2 us after issue, a K-tile computes in ~1 us: one tile in
    // flight leaves the matrix pipe waiting for data.
    constexpr int PRE = NSLOT - 1;
    constexpr int KSTEPS = BKA / 16;
    const int nk = MODCR_DBG(p.debug & 2) ? 0 : (H / BKA);
    auto gemm_loop = [&](auto FG) {
        constexpr int fgc = decltype(FG)::value;
#pragma unroll
        for (int t = 0; t < PRE; ++t)
            if (t < nk) stage(t, t * BKA);
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 1 - kt;                    // tiles issued after kt
            if (rem >= PRE - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRE - 1) * CPW) : "memory");
            else if (PRE >= 3 && rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + PRE < nk && !MODCR_DBG(p.debug & 4)) stage((kt + PRE) % NSLOT, (kt + PRE) * BKA);   // debug bit2: timing-only, no refill
            const unsigned char* sXs = smem + (kt % NSLOT) * STAGE;
            const unsigned char* sWs = sXs + (XCH + WCHH * hd) * 1024;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                bf16x8 fx[2], fw[3];
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    fx[t] = *reinterpret_cast<const bf16x8*>(sXs + soff((2 * tg + t) * 32 + r, ks * 2 + h));
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    fw[j] = *reinterpret_cast<const bf16x8*>(sWs + soff((3 * fgc + j) * 32 + r, ks * 2 + h));
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        if (fgc == 0 || j == 0)   // Q, K: features in registers, tokens on lanes
                            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[j], fx[t], acc[t][j], 0, 0, 0);
                        else                      // V: tokens in registers, features on lanes
                            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fx[t], fw[j], acc[t][j], 0, 0, 0);
                    }
            }
        }
    };
    if (fg == 0) gemm_loop(std::integral_constant<int, 0>{});
    else gemm_loop(std::integral_constant<int, 1>{});
    __syncthreads();        // every wave is done with the staging ring before it becomes the Q/K/V images

    // ---- write Q (scaled by 1/8), K, V^T (+bias) as bf16 images into LDS ----------------------
    auto put_feat = [&](const f32x16& v16, int trow, int fbase, bool isq) {
        // features fbase + 8g + 4h + e in the registers, token row `trow` on this lane
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f0 = fbase + 8 * g + 4 * h;
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v16[4 * g + e] + sBias[(isq ? 0 : 64) + f0 + e];
                if (isq) t *= 0.125f * LOG2E;
                v[e] = (bf16)t;
            }
            if (isq) {
                const int qi = trow - P;
                if (qi >= 0) *reinterpret_cast<bf16x4*>(sQ + swz128(qi, f0 >> 3) + (f0 & 7) * 2) = v;
            } else {
                *reinterpret_cast<bf16x4*>(sK + swz128(trow, f0 >> 3) + (f0 & 7) * 2) = v;
            }
        }
    };
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tb = (2 * tg + t) * 32;
        if (fg == 0) {
            put_feat(acc[t][0], tb + r, 0, true);
            put_feat(acc[t][1], tb + r, 32, true);
            put_feat(acc[t][2], tb + r, 0, false);
        } else {
            put_feat(acc[t][0], tb + r, 32, false);
#pragma unroll
            for (int j = 1; j < 3; ++j) {
                const int f = (j - 1) * 32 + r;
                const float bv = sBias[128 + f];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int t0 = tb + 8 * g + 4 * h;       // first of 4 consecutive tokens
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (bf16)(acc[t][j][4 * g + e] + bv);
                    *reinterpret_cast<bf16x4*>(sVt + f * VT_STRIDE + t0 * 2) = v;
                }
            }
        }
    }
    __syncthreads();

    // ---- chunk-mean query (phase-3 layers of seq_enc) ----------------------------------------
    if (p.chunk_id) {
        const int T = p.chunk_t;
        bf16x4 mean[8];
        bool have[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = ltid + NTH * it;
            const int t = item >> 4, c4 = item & 15;
            have[it] = false;
            if (t < T) {
                const int id = sCid[t];
                if (id >= 0) {
                    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                    int cnt = 0;
                    for (int u = 0; u < T; ++u) {
                        if (sCid[u] == id) {
                            const bf16x4 q = *reinterpret_cast<const bf16x4*>(sQ + swz128(u, c4 >> 1) + (c4 & 1) * 8);
                            s0 += (float)q[0]; s1 += (float)q[1]; s2 += (float)q[2]; s3 += (float)q[3];
                            ++cnt;
                        }
                    }
                    const float inv = 1.0f / (float)cnt;
                    mean[it][0] = (bf16)(s0 * inv); mean[it][1] = (bf16)(s1 * inv);
                    mean[it][2] = (bf16)(s2 * inv); mean[it][3] = (bf16)(s3 * inv);
                    have[it] = true;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int item = ltid + NTH * it;
            const int t = item >> 4, c4 = item & 15;
            if (have[it]) *reinterpret_cast<bf16x4*>(sQ + swz128(t, c4 >> 1) + (c4 & 1) * 8) = mean[it];
        }
        __syncthreads();
    }

    if (MODCR_DBG(p.debug & 1)) return;
    // ---- phase B: per 32-key tile S^T = K.Q^T -> online softmax -> O^T += V^T.P^T ---------------
    // Keys of a tile sit in the 16 accumulator registers (key = 32kt + (e&3) + 8(e>>2) + 4h), the
    // query on the lane: the row max / sum are register reductions plus one exchange with lane^32,
    // and the exponentiated tile, converted pairwise to bf16, is the B operand of the P.V MFMA.
    const int q0 = wave * 32;
    const int qi = q0 + r;                                  // this lane's query
    const int LW = (L + 31) >> 5;
    bf16x8 fq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        fq[ks] = *reinterpret_cast<const bf16x8*>(sQ + swz128(qi, ks * 2 + h));

    auto score_tile = [&](int kt, f32x16& sc) {             // masked log2-domain scores of key tile kt
        // the accumulator starts from the additive mask (0 / -10000*log2e / -inf for tile padding), so
        // masking costs no VALU pass after the MFMAs
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 mk = *reinterpret_cast<const f32x4*>(sMask + kt * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[4 * g + e] = mk[e];
        }
        if (p.bits) {                                       // dense mask: bit (8g+4h+e) of this query's word
            const uint32_t word = (qi < S && kt * 32 < L) ? p.bits[((int64_t)n * S + qi) * LW + kt] : 0xffffffffu;
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (!((word >> (8 * (e >> 2) + 4 * h + (e & 3))) & 1u)) sc[e] += MODCR_NEG * LOG2E;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 fk = *reinterpret_cast<const bf16x8*>(sK + swz128(kt * 32 + r, ks * 2 + h));
            sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, fq[ks], sc, 0, 0, 0);
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NW; ++kt) {
        if (kt * 32 >= L) break;                            // block-uniform: tile holds padding only
        f32x16 sc;
        score_tile(kt, sc);
        float tmax = sc[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) tmax = fmaxf(tmax, sc[e]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);             // finite: tile 0 always holds key 0 < L
        float lsum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float ex = __builtin_amdgcn_exp2f(sc[e] - m_new);
            sc[e] = ex;
            lsum += ex;
        }
        if (__any(m_new > m_run)) {                         // wave-uniform: some row's max moved -> rescale
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
        }
        l_run += lsum;
        m_run = m_new;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (bf16)sc[8 * s2 + j];
            if (p.drop_on) {
                // attention-probability dropout (training mode; the row sum above took the unmasked weights): registers
                // 0-3 / 4-7 of pb are keys 32 kt + 16 s2 + 4 h + 0..3 and + 8, i.e. the l4 groups h and h + 2 of the 16-key
                // block 2 kt + s2 of this lane's query (mask layout: attn_common.h)
                u32x4_t w = __builtin_bit_cast(u32x4_t, pb);
                const uint32_t row = (uint32_t)((n * p.A + a) * 256 + qi);
                const uint32_t b0 = attn_drop_base(row, (uint32_t)h, p.drop_key), b1 = attn_drop_base(row, (uint32_t)(h + 2), p.drop_key);
                const int j0 = (2 * kt + s2) * 2;
                w[0] &= attn_keep2(attn_drop_word(b0, j0, p.drop_key), p.drop_thr2);
                w[1] &= attn_keep2(attn_drop_word(b0, j0 + 1, p.drop_key), p.drop_thr2);
                w[2] &= attn_keep2(attn_drop_word(b1, j0, p.drop_key), p.drop_thr2);
                w[3] &= attn_keep2(attn_drop_word(b1, j0 + 1, p.drop_key), p.drop_thr2);
                pb = __builtin_bit_cast(bf16x8, w);
            }
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const unsigned char* vrow = sVt + (dt * 32 + r) * VT_STRIDE + (kt * 32 + 16 * s2 + 4 * h) * 2;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + 16);
                bf16x8 va;
                va[0] = lo[0]; va[1] = lo[1]; va[2] = lo[2]; va[3] = lo[3];
                va[4] = hi[0]; va[5] = hi[1]; va[6] = hi[2]; va[7] = hi[3];
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, o[dt], 0, 0, 0);
            }
        }
    }
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
    const float inv_ctx = p.drop_on ? inv * p.drop_keep : inv;              // context rows carry dropout's 1 / (1 - p)

    // optional side outputs (parity tests, align-loss layers): recompute each tile against the
    // final row max / sum
    if (p.probs || p.align_map) {
        float* pr = (p.probs && qi < S) ? p.probs + (((int64_t)n * p.A + a) * S + qi) * L : nullptr;
        const bool amap = p.align_map && q0 < p.align_t;                 // wave-uniform
        const int T = p.align_t, R = S - T;
        float* sS = reinterpret_cast<float*>(sQ + q0 * 128);             // this wave's own 4 KB (Q rows are in registers)
#pragma unroll
        for (int kt = 0; kt < NW; ++kt) {
            if (kt * 32 >= L) break;
            f32x16 sc;
            score_tile(kt, sc);
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[e] = __builtin_amdgcn_exp2f(sc[e] - m_run) * inv;
            if (pr) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (key < L) pr[key] = sc[e];
                }
            }
            if (amap && kt * 32 + 31 >= P + T) {                          // tile holds region keys
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = (e & 3) + 8 * (e >> 2) + 4 * h;
                    sS[r * 32 + (c ^ r)] = sc[e];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int row = it * 2 + h, col = r;
                    const int key = kt * 32 + col, q = q0 + row;
                    const float v = sS[row * 32 + (col ^ row)];
                    if (q < T && key >= P + T && key < L)
                        atomicAdd(p.align_map + ((int64_t)n * T + q) * R + (key - P - T), v);
                }
            }
        }
    }

    // epilogue: normalise, transpose through this wave's private 4 KB, store 128-byte rows
    {
        unsigned char* sO = sQ + q0 * 128;       // rows q0..q0+31, 128 B each, plain (unswizzled)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (bf16)(o[dt][4 * g + e] * inv_ctx);
                const int d0 = dt * 32 + 8 * g + 4 * h;
                *reinterpret_cast<bf16x4*>(sO + r * 128 + ((((d0 >> 3) ^ r) & 7) << 4) + (d0 & 7) * 2) = v;
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), ch = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(sO + row * 128 + (((ch ^ row) & 7) << 4));
            const int q = q0 + row;
            if (q < S)
                *reinterpret_cast<uint4*>(p.ctx + ((int64_t)n * S + q) * H + a * 64 + ch * 8) = v;
        }
    }
}

template <int NW, int HPW, int BKA, int NSLOT> constexpr size_t attn_smem_bytes() {
    constexpr int LP = NW * 32;
    constexpr size_t end_a = (size_t)NSLOT * (LP + 192 * HPW) * BKA * 2;
    constexpr size_t end_b = (size_t)HPW * (2 * LP * 128 + 64 * (LP * 2 + VT_PAD));
    return (end_a > end_b ? end_a : end_b) + (size_t)LP * 4 + 192 * HPW * 4 + (size_t)LP * 4;
}

template <int NW, int HPW, int OCC, int BKA, int NSLOT>
int launch_attn(const AttnArgs& p, hipStream_t st) {
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];   // idempotent attribute set; benign if raced
    const size_t smem = attn_smem_bytes<NW, HPW, BKA, NSLOT>();
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_attn_bf16_kernel<NW, HPW, OCC, BKA, NSLOT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) {
            modcr_set_error("qkv_attn: cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    launch_timed(qkv_attn_bf16_kernel<NW, HPW, OCC, BKA, NSLOT>, dim3(p.N * (p.A / HPW)), dim3(NW * HPW * 64), smem, st, p);
    return modcr_check_launch("qkv_attn_bf16");
}

// ---- v4: 8 waves in two staggered groups, half-tile ring (the GEMM's 8-phase schedule) ----------
// One workgroup = one sequence x two heads, 128 < S <= 192, no prefix.
//   phase A  [192 tokens] x [384 features = (q|k|v) of two heads] over K = H on the schedule of
//            linear_bf16_p8_kernel (gemm.hip): K-tiles of 64 split into four half-tiles {A0, B0, B1, A1}
//            (A = 96 token rows = 12 KB: rows wr*48.. of both wave rows; B = 192 feature rows = 24 KB),
//            one half-tile staged per phase six phases ahead by LDS-DMA, counted vmcnt that leaves four
//            half-tiles (10 DMA instructions per wave) in flight, 18 v_mfma_f32_16x16x32_bf16 between
//            two raw barriers; the wave groups wr = 0 / 1 (one wave of each per SIMD) run one barrier
//            apart.  Wave (wr, wc) owns tokens wr*96.. x {64 features of q or k of head wc>>1, 32 features
//            of v of head wc>>1}: q/k blocks are produced with the weights as the MFMA A operand
//            (4 consecutive features per lane -> 8-byte stores into the row-major [token][64] images),
//            v blocks the other way round ([64][token] image).
//   phase B  wave = (head, 48 queries).  S^T = K.Q^T for all 192 keys stays in registers (the 144
//            accumulator registers of phase A), the additive mask is the MFMA accumulator init, so the
//            row max is exact: one pass, no rescale, two cross-lane steps per 16 queries in total.
//            P = exp2(S - max) in place, bf16 pairs of two 16-key blocks are the B operand of
//            O^T += V^T.P^T (keys permuted identically on the V^T fragment reads).
// NH = heads per workgroup: 2 for the token tiles 128 and 192 (64 < S <= 192); 1 for the 256-token tile (192 < S <= 256, the
// VCR / Oscar-large shape class S = 230): two heads' Q | K | V^T images of 256 tokens do not fit the 160 KB of LDS.  The 8 waves
// of phase A form WR wave rows x WC wave columns (a wave column = 96 features: q or k of a head + half of its v), i.e.
// 2 x 4 (NH = 2) or 4 x 2 (NH = 1); phase B gives every head 8 / NH waves of QW queries each.
template <int LP_, int NH_ = 2>
struct A4T {
    static constexpr int LP = LP_, NH = NH_, NF = 192 * NH, NT = 512;   // token tile, heads, features (q|k|v of NH heads)
    static constexpr int WC = 2 * NH, WR = 8 / WC;              // wave grid of phase A
    static constexpr int RW = LP / WR;                          // tokens per wave row
    static constexpr int QW = RW / 2;                           // tokens per (wave row, half) = queries per wave in phase B
    static constexpr int NI = QW / 16;                          // 16-token blocks per (wave row, half)
    static constexpr int NQB = QW / 16, NKT = LP / 32;          // query blocks per wave, 32-key tiles
    static constexpr int WPH = 8 / NH;                          // waves per head in phase B
    static constexpr int NA = (LP / 2 > 64) ? 2 : 1;            // LDS-DMA instructions per wave per A half (LP = 192: the 2nd by lanes 0..31)
    static constexpr bool A_HALF_PIECE = (LP / 2) % 64 != 0;    // the 2nd A instruction moves half a piece (4 rows per wave)
    static constexpr int NBI = NH == 2 ? 3 : 2;                 // ... per B half (NH = 1: one piece + half a piece by lanes 0..31)
    static constexpr int HA = (LP / 2) * 128, HB = (NF / 2) * 128;   // bytes per half-tile
    static constexpr int KT = 2 * HA + 2 * HB;                  // A0 | A1 | B0 | B1
    static constexpr int RING = 2 * KT;
    static constexpr int VT_STRIDE = LP * 2 + VT_PAD;
    static constexpr int IMGS = 2 * NH * LP * 128 + NH * 64 * VT_STRIDE;
    static constexpr int MAIN = (IMGS > RING) ? IMGS : RING;
    static constexpr int DROP_OFF = MAIN + LP * 4 + NF * 4 + LP * 4 + 16 + 3 * LP * 4;   // 8 dwords: attention-dropout parameters, P
    static constexpr int SMEM = DROP_OFF + 32;
    static constexpr int FOLD_OFF = SMEM;                       // FOLD variants: sRs[LP] | sRm[LP] | sC[NF] floats behind everything else
    static constexpr int SMEM_FOLD = FOLD_OFF + 2 * LP * 4 + NF * 4;
    static constexpr int FLY4 = 2 * NA + 2 * NBI;               // DMA instructions per wave in four consecutive half-tiles
    // phase-B images over the ring: [Q0 | Q1 | K0 | K1 | Vt0 | Vt1]  (NH = 1: [Q | K | Vt])
    static __device__ __forceinline__ unsigned char* img_qk(unsigned char* smem, int part, int head) { return smem + (part * NH + head) * LP * 128; }
    static __device__ __forceinline__ unsigned char* img_vt(unsigned char* smem, int head) { return smem + 2 * NH * LP * 128 + head * 64 * VT_STRIDE; }
};
typedef A4T<192> A4;

// epilogue of one wave: O^T / rowsum -> bf16 -> transposed through the wave's own 48 Q rows -> 128-byte row stores
// rows [row_lo, row_hi) of the wave's 48 / 32 tile rows are query rows of the sequence (prefix rows in front and the padded
// tail behind are not); ctx_rows points at the output row of tile row 0 of the wave (possibly before the sequence: masked)
template <typename A4>
__device__ __forceinline__ void attn4_store_ctx(const f32x4 (&o)[4][A4::NQB], const float (&inv)[A4::NQB], unsigned char* sO, bf16* ctx_rows,
                                                int H, int row_lo, int row_hi, int l15, int l4, int lane) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
    for (int qb = 0; qb < A4::NQB; ++qb)
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (bf16)(o[db][qb][e] * inv[qb]);
            const int row = qb * 16 + l15, d0 = db * 16 + 4 * l4;
            *reinterpret_cast<bf16x4*>(sO + row * 128 + ((((d0 >> 3) ^ row) & 7) << 4) + (d0 & 7) * 2) = v;
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
    for (int it = 0; it < A4::QW / 8; ++it) {
        const int row = it * 8 + (lane >> 3), ch = lane & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(sO + row * 128 + (((ch ^ row) & 7) << 4));
        if (row >= row_lo && row < row_hi) *reinterpret_cast<uint4*>(ctx_rows + (int64_t)row * H + ch * 8) = v;
    }
}

// Chunk-mean queries (v10:66-78) on the Q images of both heads, for the whole workgroup (contains barriers).
// Out of line: its twelve unrolled items would otherwise shape the register allocation of the kernel around it.
template <int LP, int NH>
__device__ __attribute__((noinline)) void attn4_chunk_mean(unsigned char* smem, int T, int tid) {
    typedef A4T<LP, NH> A4;                                 // NH = 2: 256 threads per head; NH = 1: all 512 on the one head
    constexpr int TPH = A4::NT / NH, ITS = LP * 16 / TPH;
    float* sMask = reinterpret_cast<float*>(smem + A4::MAIN);
    int* sCid = reinterpret_cast<int*>(sMask + LP + A4::NF);
    int* sFlag = sCid + LP;
    int* sFirst = sFlag + 4;
    int* sLast = sFirst + LP;
    int* sCnt = sLast + LP;
    unsigned char* sQ = A4::img_qk(smem, 0, tid / TPH);
    const int ltid = tid % TPH;
    // first / last token and size of every chunk id; when each chunk is one contiguous run of tokens (the
    // data format: utils/GetChunk_v4_vcr.py offsets are consecutive token indices) a token's mean is a sum over
    // [first, last] instead of a scan of all T tokens.  Anything else takes the scan.
    if (tid < T) {
        const int id = sCid[tid];
        if (id >= LP) sFlag[1] = 1;
        else if (id >= 0) { atomicMin(&sFirst[id], tid); atomicMax(&sLast[id], tid); atomicAdd(&sCnt[id], 1); }
    }
    __syncthreads();
    if (tid < LP && sCnt[tid] > 0 && sLast[tid] - sFirst[tid] + 1 != sCnt[tid]) sFlag[1] = 1;
    __syncthreads();
    const bool runs = sFlag[1] == 0;
    bf16x4 mean[ITS];
    bool have[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int item = ltid + TPH * it;
        const int t = item >> 4, c4 = item & 15;
        have[it] = false;
        if (t < T) {
            const int id = sCid[t];
            if (id >= 0) {
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int cnt = 0;
                const int u0 = runs ? sFirst[id] : 0, u1 = runs ? sLast[id] + 1 : T;
                for (int u = u0; u < u1; ++u) {
                    if (sCid[u] == id) {
                        const bf16x4 q = *reinterpret_cast<const bf16x4*>(sQ + swz128(u, c4 >> 1) + (c4 & 1) * 8);
                        s0 += (float)q[0]; s1 += (float)q[1]; s2 += (float)q[2]; s3 += (float)q[3];
                        ++cnt;
                    }
                }
                const float inv = 1.0f / (float)cnt;
                mean[it][0] = (bf16)(s0 * inv); mean[it][1] = (bf16)(s1 * inv);
                mean[it][2] = (bf16)(s2 * inv); mean[it][3] = (bf16)(s3 * inv);
                have[it] = true;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int item = ltid + TPH * it;
        const int t = item >> 4, c4 = item & 15;
        if (have[it]) *reinterpret_cast<bf16x4*>(sQ + swz128(t, c4 >> 1) + (c4 & 1) * 8) = mean[it];
    }
    __syncthreads();
}

// Exact phase B of one tile for the whole workgroup (all 8 waves call it together: it contains barriers): all 192
// scores of a row in registers, exact row max, optional probabilities / head-summed text->region map, context
// store.  The generic variant (MODE 0) always runs it; the production variants only when the streaming pass
// found a row sum out of range, so it is kept out of line (its 144 score registers would otherwise shape the
// register allocation of the hot path).
template <int LP, int NH>
__device__ __attribute__((noinline)) void attn4_exact_tail(unsigned char* smem, const uint32_t* bits, float* probs, float* align_map,
                                                           bf16* ctx, int align_t, int S, int H, int A, int n, int a0, int tid) {
    // (prefix rows: P rides in the LDS parameter block, see sDrop[4]; S = query rows, L = P + S keys / tile rows)
    typedef A4T<LP, NH> A4;
    // attention-dropout parameters: left in LDS by the kernel (more call arguments change how the CALLER's accumulators
    // are kept around the call: measured 100 MB of scratch traffic per launch on the common path)
    const uint32_t* sDrop = reinterpret_cast<const uint32_t*>(smem + A4::DROP_OFF);
    const uint32_t drop_thr2 = sDrop[0];
    const uint64_t drop_key = ((uint64_t)sDrop[2] << 32) | sDrop[1];
    const float drop_keep = __uint_as_float(sDrop[3]);
    const bool drop_on = drop_keep != 1.0f;                 // the kernel leaves 1.0 there when the masking is off
    const int P = (int)sDrop[4];
    float* lse_out = reinterpret_cast<float*>(((uint64_t)sDrop[6] << 32) | sDrop[5]);
    const bool side_post = drop_on && sDrop[7] != 0;        // probabilities / align map after the dropout (reference semantics)
    const int thr_s = (int)(short)(drop_thr2 & 0xffffu) + 1;   // signed 16-bit threshold: kept iff field >= thr_s
    constexpr int VT_STRIDE = A4::VT_STRIDE, NKT = A4::NKT, NQB = A4::NQB;
    const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int hd = wave / A4::WPH, qbase = (wave % A4::WPH) * A4::QW, a = a0 + hd, L = P + S;
    const int LW = (L + 31) >> 5;
    const float* sMask = reinterpret_cast<const float*>(smem + A4::MAIN);
    const unsigned char* sQ = A4::img_qk(smem, 0, hd);
    const unsigned char* sK = A4::img_qk(smem, 1, hd);
    const unsigned char* sVt = A4::img_vt(smem, hd);
    f32x4 sc[NKT][NQB][2];
    {
        bf16x8 fq[NQB][2];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fq[qb][ks] = *reinterpret_cast<const bf16x8*>(sQ + swz128(qbase + qb * 16 + l15, ks * 4 + l4));
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int krow = kt * 32 + kb * 16;
                const bf16x8 fk0 = *reinterpret_cast<const bf16x8*>(sK + swz128(krow + l15, l4));
                const bf16x8 fk1 = *reinterpret_cast<const bf16x8*>(sK + swz128(krow + l15, 4 + l4));
                const f32x4 mk = *reinterpret_cast<const f32x4*>(sMask + krow + 4 * l4);
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    f32x4 c = mk;
                    if (bits) {                             // dense mask: bit (16 kb + 4 l4 + e) of this query's word
                        const int qi = qbase + qb * 16 + l15 - P;
                        const uint32_t word = (qi >= 0 && qi < S && kt < LW) ? bits[((int64_t)n * S + qi) * LW + kt] : 0xffffffffu;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (!((word >> (kb * 16 + 4 * l4 + e)) & 1u)) c[e] += MODCR_NEG * LOG2E;
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk0, fq[qb][0], c, 0, 0, 0);
                    sc[kt][qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk1, fq[qb][1], c, 0, 0, 0);
                }
            }
    }
    uint32_t dbase[NQB];                                    // dropout bases of this lane's query rows (attn_common.h)
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) dbase[qb] = attn_drop_base((uint32_t)((n * A + a) * 256 + qbase + qb * 16 + l15 - P), (uint32_t)l4, drop_key);
    float mx[NQB], ls[NQB], inv[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = fmaxf(m, sc[kt][qb][kb][e]);
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));                // finite: key 0 < L
        mx[qb] = m;
        ls[qb] = 0.f;
    }
    f32x4 o[4][NQB];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        bf16x8 va[4];
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const unsigned char* vrow = sVt + (db * 16 + l15) * VT_STRIDE + (kt * 32 + 4 * l4) * 2;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow);
            const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + 32);
            va[db][0] = lo[0]; va[db][1] = lo[1]; va[db][2] = lo[2]; va[db][3] = lo[3];
            va[db][4] = hi[0]; va[db][5] = hi[1]; va[db][6] = hi[2]; va[db][7] = hi[3];
        }
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            bf16x8 pb;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ex = __builtin_amdgcn_exp2f(sc[kt][qb][kb][e] - mx[qb]);
                    sc[kt][qb][kb][e] = ex;
                    ls[qb] += ex;
                    pb[4 * kb + e] = (bf16)ex;
                }
            if (drop_on)        // the context uses the masked weights, the row sum the unmasked ones
                pb = attn_drop8_rt(pb, dbase[qb], kt, drop_key, drop_thr2);
            if (side_post) {    // side outputs after the dropout: the scores kept for them take the mask (x 1 / (1 - p) below)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t w = attn_drop_word(dbase[qb], 4 * kt + i, drop_key);
#pragma unroll
                    for (int f = 0; f < 2; ++f)
                        if ((int)(short)(w >> (16 * f)) < thr_s) sc[kt][qb][i >> 1][2 * (i & 1) + f] = 0.f;
                }
            }
#pragma unroll
            for (int db = 0; db < 4; ++db)
                o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[db], pb, o[db][qb], 0, 0, 0);
        }
    }
    float inv_ctx[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        float l = ls[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        inv[qb] = 1.0f / l;
        inv_ctx[qb] = drop_on ? inv[qb] * drop_keep : inv[qb];
        if (side_post) inv[qb] = inv_ctx[qb];               // (inv is used by the side outputs only from here on)
        if (lse_out) {                                      // log2 sum_j exp2(score_ij) = row max + log2 l
            const int qi = qbase + qb * 16 + l15 - P;
            if (l4 == 0 && qi >= 0 && qi < S) lse_out[((int64_t)n * A + a) * S + qi] = mx[qb] + __builtin_amdgcn_logf(l);
        }
    }
    // ---- side outputs: full probabilities (parity tests); head-summed text -> region block --------------
    if (probs) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const int qi = qbase + qb * 16 + l15 - P;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = kt * 32 + kb * 16 + 4 * l4 + e;
                        if (qi >= 0 && qi < S && key < L)
                            probs[(((int64_t)n * A + a) * S + qi) * L + key] = sc[kt][qb][kb][e] * inv[qb];
                    }
            }
    }
    if (align_map) {
        // both heads add their normalised text->region block into one LDS tile [T][R] (over the V^T images, dead
        // once every wave is past its P.V), then whole rows go out as atomics
        // (one head per workgroup: the tile starts at the K image, which is dead as well -- K | V^T together hold any [T][R] with
        // T + R <= 256; the context transposes below go through the Q image)
        const int T = align_t, R = S - T;
        static_assert(NH == 2 || LP * 128 + 64 * VT_STRIDE >= (LP / 2) * (LP / 2) * 4, "align tile over K | V^T");
        float* sAm = reinterpret_cast<float*>(NH == 1 ? A4::img_qk(smem, 1, 0) : A4::img_vt(smem, 0));
        __syncthreads();
        for (int j = tid; j < T * R; j += A4::NT) sAm[j] = 0.f;
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            if (kt * 32 + 31 >= T) {
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    const int qi = qbase + qb * 16 + l15;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int key = kt * 32 + kb * 16 + 4 * l4 + e;
                            if (qi < T && key >= T && key < L)
                                atomicAdd(sAm + qi * R + (key - T), sc[kt][qb][kb][e] * inv[qb]);
                        }
                }
            }
        __syncthreads();
        float* dst = align_map + (int64_t)n * T * R;
        for (int j = tid; j < T * R; j += A4::NT) atomicAdd(dst + j, sAm[j]);
    }
    attn4_store_ctx<A4>(o, inv_ctx, smem + (hd * LP + qbase) * 128, ctx + ((int64_t)n * S + qbase - P) * H + a * 64, H, P - qbase, L - qbase, l15, l4, lane);
}

// MODE 0 = generic (probabilities / align map / chunk-mean queries / either mask form, exact row max), one tile
// per workgroup.  Production variants (persistent over tiles): 1 = broadcast key mask, 2 = dense mask bits,
// 3 = dense mask bits + chunk-mean queries + head-summed text->region map (seq_enc layers 9-11): streaming
// softmax without a max pass, row sums checked and the tile redone exactly when one leaves [1e-30, 1e30].
// DUMPV = 1 (two heads per workgroup, streaming variants, p.dump set): the Q | K | V image dump of a trainable layer's forward is
// issued in 18 (12 at LP = 128) pieces per thread BETWEEN the key tiles of phase B instead of as a block in front of it: the dump is
// store-issue work (144 KB per tile), phase B is VALU / MFMA work, and the block cost +58-75 us per call.
// (FOLD, the LayerNorm-fold prototype of DESIGN section 7, is a sixth template parameter in the TUNING build only: the product's
// kernels keep the five-parameter names and the binaries that the profiles/ and PMC files of the round were taken on)
#ifdef MODCR_TUNING
#define A4_FOLD_PARAM , int FOLD = 0
#define A4_FOLD_ARG , FOLD
#else
#define A4_FOLD_PARAM
#define A4_FOLD_ARG
#endif
template <int KMODE, int LP, int DROP, int NHD = 2, int DUMPV = 0 A4_FOLD_PARAM>
__global__ __launch_bounds__(512, 2) void qkv_attn4_kernel(AttnArgs p) {
#ifndef MODCR_TUNING
    constexpr int FOLD = 0;
#endif
    static_assert(!DUMPV || (NHD == 2 && KMODE != 0), "interleaved dump: streaming variants with two heads per workgroup");
    typedef A4T<LP, NHD> A4;
    constexpr int NI = A4::NI, QW = A4::QW, NQB = A4::NQB, NKT = A4::NKT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HA = A4::HA, HB = A4::HB, KT = A4::KT, VT_STRIDE = A4::VT_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / A4::WC, wc = wave % A4::WC;
    const int l15 = lane & 15, l4 = lane >> 4;
    // tuning build, A/B only: static priority for one half of the waves for the whole kernel (MI355X_MICROARCH.md "Two waves per SIMD"
    // item 4); debug bit 8 raises waves 4-7 (the younger half), bit 9 waves 0-3 (the control)
    if (MODCR_DBG(p.debug & 256) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (MODCR_DBG(p.debug & 512) && wave < 4) __builtin_amdgcn_s_setprio(1);
    float* sMask = reinterpret_cast<float*>(smem + A4::MAIN);
    float* sBias = sMask + LP;                              // [head][q|k|v][64]
    int* sCid = reinterpret_cast<int*>(sBias + A4::NF);
    int* sFlag = sCid + LP;                                 // [0] redo the tile with the exact pass, [1] chunks are not contiguous runs
    int* sFirst = sFlag + 4;                                // per chunk id: first / last token, token count
    int* sLast = sFirst + LP;
    int* sCnt = sLast + LP;

    const int hgroups = p.A / NHD;
    const int ntiles = p.N * hgroups;
    // prefix rows (history_state, modeling_bert.py:36-44): p.x then holds [prefix ; x] rows per sequence (the host entry
    // concatenates them), tile rows / keys 0..L-1 with L = P + S, query rows = tile rows P..L-1 -> output rows 0..S-1
    const int S = p.S, P = p.P, L = p.S + p.P, H = p.H;
    int n = 0, a0 = 0;                                      // sequence and first head of the current tile

    // ---- DMA sources: uniform base (sequence / weight matrix + k offset) + 32-bit per-lane byte offset.
    // A half mh: LDS row r (0..95) = token (r / 48) * 96 + mh * 48 + r % 48; piece `wave` (8 rows) by all
    // 64 lanes, rows 64 + 4 wave.. by lanes 0..31 (the LDS address of an LDS-DMA is base + 16 lane).
    // B half nh: LDS row r (0..191) = wave column r / 48, feature slot cc = r % 48 of that column.
    unsigned offA[2][2], vB;
    // B pieces hold 8 consecutive weight rows (a piece never straddles a 16-feature block), so the source is
    // a per-piece scalar row (wbrow, relative to head a0's rows) + ONE per-lane offset: row (lane / 8) and the
    // swizzled chunk, whose key (r >> 1) & 7 = (4 piece + lane / 16) & 7 depends on the piece's parity = wave & 1.
    // (NH = 1: a B half is 96 rows = one piece per wave + half a piece: rows 64 + 4 wave.. by lanes 0..31 with their own
    // per-lane offset vB2, whose swizzle key (r >> 1) & 7 = (2 wave + lane / 16) & 7 depends on wave & 3)
    int wbrow[2][3];
    unsigned vB2 = 0;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int q = 0; q < A4::NBI; ++q) {
            const int r = (NHD == 1 && q == 1) ? 64 + 4 * wave : 8 * (wave + 8 * q);
            const int wcr = r / 48, cc = r % 48;
            int part, d;
            if (nh == 0) { part = wcr & 1; d = cc; }
            else if (cc < 16) { part = wcr & 1; d = 48 + cc; }
            else { part = 2; d = (wcr & 1) * 32 + cc - 16; }
            wbrow[nh][q] = part * H + (wcr >> 1) * 64 + d;
        }
    auto uniform_ptr = [](const void* q) {
        const uint64_t b64 = reinterpret_cast<uint64_t>(q);
        return reinterpret_cast<const char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                                             (unsigned)__builtin_amdgcn_readfirstlane((int)(b64 & 0xffffffffu)));
    };
    const bf16* xb = p.x;                                   // x of sequence n / Wqkv from head a0's rows on (per tile)
    const bf16* wt = p.wqkv;
    // kind: 0 = A0, 1 = B0, 2 = B1, 3 = A1 (staging order); LDS order inside a K-tile buffer: A0 A1 B0 B1
    auto stage_half = [&](int buf, int kind, int k0) {
        if (kind == 0 || kind == 3) {
            const int mh = kind == 3;
            unsigned char* dst = smem + buf * KT + mh * HA;
            const char* base = uniform_ptr(xb + k0);
            glds16(offA[mh][0], base, dst + wave * 1024);
            if constexpr (A4::NA == 2 && A4::A_HALF_PIECE) { if (lane < 32) glds16(offA[mh][1], base, dst + 8192 + wave * 512); }
            if constexpr (A4::NA == 2 && !A4::A_HALF_PIECE) glds16(offA[mh][1], base, dst + 8192 + wave * 1024);
        } else {
            const int nh = kind == 2;
            unsigned char* dst = smem + buf * KT + 2 * HA + nh * HB;
            if constexpr (NHD == 2) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    glds16(vB, uniform_ptr(wt + (int64_t)wbrow[nh][q] * H + k0), dst + (wave + 8 * q) * 1024);
            } else {
                glds16(vB, uniform_ptr(wt + (int64_t)wbrow[nh][0] * H + k0), dst + wave * 1024);
                if (lane < 32) glds16(vB2, uniform_ptr(wt + (int64_t)wbrow[nh][1] * H + k0), dst + 8192 + wave * 512);
            }
        }
    };

    typedef const __attribute__((address_space(3))) bf16x8* lds_v8;
    unsigned aA[2][2], aB[2][2];

    f32x4 acc[2][2][NI][3];
    bf16x8 fa[NI][2], fb[2][3][2];
    auto rdA = [&](int buf, int mh) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            fa[i][0] = *(lds_v8)(aA[buf][0] + mh * HA + i * 2048);
            fa[i][1] = *(lds_v8)(aA[buf][1] + mh * HA + i * 2048);
        }
    };
    auto rdB = [&](int buf, int nh) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            fb[nh][j][0] = *(lds_v8)(aB[buf][0] + nh * HB + j * 2048);
            fb[nh][j][1] = *(lds_v8)(aB[buf][1] + nh * HB + j * 2048);
        }
    };
    const int nk = MODCR_DBG(p.debug & 2) ? 4 : (H >> 6);       // K-tiles (even, >= 4); debug bit 1: timing-only short loop
    // (128-token tile only: the other tiles' K loops are compiled as before)
    constexpr bool SKIP_PAD = (LP == 128);
    [[maybe_unused]] const bool skip_last = SKIP_PAD && (wr * A4::RW + QW + (NI - 1) * 16 >= p.S + p.P) && !MODCR_DBG(p.debug & 4096);
    // phase I of an 8-phase trip (two K-tiles).  KMODE 0 = steady state, 1 = last trip.
    auto phase = [&](auto I_, auto MODE_, int kt) {
        constexpr int I = decltype(I_)::value, MODE = decltype(MODE_)::value;
        constexpr int Q = I & 3, BUF = I >> 2;
        constexpr int MH = (Q >= 2), NH = (Q == 1 || Q == 2);
        if constexpr (Q == 0) { rdB(BUF, 0); __builtin_amdgcn_sched_barrier(0); rdA(BUF, 0); }
        if constexpr (Q == 1) rdB(BUF, 1);
        if constexpr (Q == 2) rdA(BUF, 1);
        if constexpr (MODE != 1 || I < 2) {
            constexpr int KIND = (I + 2) & 3, DT = (I + 6) >> 2;
            stage_half(DT & 1, KIND, (kt + DT) << 6);
        }
        // DMA instructions per wave that may stay in flight: the last min(4, remaining) half-tiles staged
        constexpr int VM = MODE != 1 ? A4::FLY4 : (I <= 1 ? A4::FLY4 : I == 2 ? A4::NA + 2 * A4::NBI : I == 3 ? A4::NA + A4::NBI : I == 4 ? A4::NA : 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
        auto block = [&](int ks, int i, int j) {
            if (NH == 1 && j >= 1)   // v: tokens in registers, features on lanes
                acc[MH][NH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb[NH][j][ks], acc[MH][NH][i][j], 0, 0, 0);
            else                     // q, k: features in registers, tokens on lanes
                acc[MH][NH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[NH][j][ks], fa[i][ks], acc[MH][NH][i][j], 0, 0, 0);
        };
        if constexpr (SKIP_PAD && MH == 1) {
            // 128-token tile (round 6): the last 16-token block of a wave row's second half (tokens 112..127 for wave row 1) is pure tile
            // padding whenever L <= 112 -- the image-only pass (S = 101) and the RoBERTa body (P + S = 111) -- and its products are the
            // only difference to a 112-token tile in this phase: skipped (its accumulators stay 0; those image rows are masked keys and
            // unstored queries).  One wave of every SIMD is a wave-row-1 wave, so the MH = 1 phases drop from 24 to 18 MFMAs per SIMD.
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NI - 1; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) block(ks, i, j);
            if (!skip_last) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int j = 0; j < 3; ++j) block(ks, NI - 1, j);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) block(ks, i, j);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto trip = [&](auto MODE_, int kt) {
        phase(std::integral_constant<int, 0>{}, MODE_, kt);
        phase(std::integral_constant<int, 1>{}, MODE_, kt);
        phase(std::integral_constant<int, 2>{}, MODE_, kt);
        phase(std::integral_constant<int, 3>{}, MODE_, kt);
        phase(std::integral_constant<int, 4>{}, MODE_, kt);
        phase(std::integral_constant<int, 5>{}, MODE_, kt);
        phase(std::integral_constant<int, 6>{}, MODE_, kt);
        phase(std::integral_constant<int, 7>{}, MODE_, kt);
    };
    // Persistent: workgroup b walks tiles b, b + gridDim, ... (gridDim is a multiple of 8, so a workgroup's
    // tiles stay in its XCD's chunk of the tile order); no workgroup launch / LDS hand-over between tiles.
    // (A sequence-major walk -- one workgroup owns a sequence's six head pairs, its align map accumulated by plain
    // read-add-write -- measured 178 vs 174 us at N = 256 and 269 vs 257 us for MODE 3: not kept.)
    [[maybe_unused]] int trace_it = 0;
    [[maybe_unused]] auto trace = [&](int ev) {
        if (MODCR_DBG(p.trace != nullptr)) {
            if (blockIdx.x == 0 && lane == 0 && (wave & 3) == 0 && trace_it < 32) p.trace[(wave >> 2) * 512 + trace_it * 8 + ev] = __builtin_readcyclecounter();
        }
    };
    for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
    trace(0);
    {
        if (p.hconc > 0 && (p.N & 7) == 0) {
            // head-pair-major inside an XCD, `hconc` head pairs at a time: the workgroups resident on an XCD share hconc weight
            // slices (hconc x 590 KB of its 4 MB L2) instead of all of Wqkv
            const int xcd = vt & 7, idx = vt >> 3, ns = p.N >> 3, G = p.hconc;
            const int hgb = idx / (ns * G), rem = idx - hgb * ns * G;
            const int gw = min(G, hgroups - hgb * G);
            n = xcd * ns + rem / gw;
            a0 = (hgb * G + rem % gw) * NHD;
        } else {
            const int tile = xcd_remap(vt, ntiles);
            n = tile / hgroups; a0 = (tile % hgroups) * NHD;
        }
        xb = p.x + (int64_t)n * L * H;
        wt = p.wqkv + (int64_t)a0 * 64 * H;
    }
    {
        // Per-tile recomputation (from an opaque copy of the thread id) of every per-lane address the K loop
        // needs: kept loop-invariant across tiles they would stay live through phase B and spill there.
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int lane = tq & 63, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = q == 0 ? 8 * wave + (lane >> 3)
                                     : (A4::A_HALF_PIECE ? 64 + 4 * wave + ((lane & 31) >> 3) : 64 + 8 * wave + (lane >> 3));
                const int tok = min((r / QW) * A4::RW + mh * QW + (r % QW), L - 1);      // padding rows re-read row L-1
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                offA[mh][q] = (unsigned)((tok * H + c * 8) * 2);
            }
        vB = (unsigned)((((lane >> 3) * H) + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8)) * 2);
        if constexpr (NHD == 1) vB2 = (unsigned)(((((lane & 31) >> 3) * H) + (((lane & 7) ^ ((2 * wave + ((lane & 31) >> 4)) & 7)) * 8)) * 2);
    const int keyr = (l15 >> 1) & 7;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
        const unsigned ck0 = ((l4 ^ keyr) & 7) << 4, ck1 = (((l4 + 4) ^ keyr) & 7) << 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            aA[b][0] = lds0 + b * KT + (wr * QW + l15) * 128 + ck0;
            aA[b][1] = lds0 + b * KT + (wr * QW + l15) * 128 + ck1;
            aB[b][0] = lds0 + b * KT + 2 * HA + (wc * 48 + l15) * 128 + ck0;
            aB[b][1] = lds0 + b * KT + 2 * HA + (wc * 48 + l15) * 128 + ck1;
            asm volatile("" : "+v"(aA[b][0]), "+v"(aA[b][1]), "+v"(aB[b][0]), "+v"(aB[b][1]));
        }

    }
    {
        int tidb = tid;                                     // opaque per tile: keeps this address math inside the loop
        asm volatile("" : "+v"(tidb));
        if (tidb < LP) {
            const int j = tidb;
            float m;
            if (j >= L) m = -INFINITY;
            else if (p.bits) m = 0.f;
            else m = (1.0f - p.key_mask[(int64_t)n * L + j]) * (MODCR_NEG * LOG2E);
            sMask[j] = m;
            sCid[j] = (p.chunk_id && j < p.chunk_t) ? p.chunk_id[(int64_t)n * p.chunk_t + j] : -1;
        }
        if (tidb == 0) {
            sFlag[0] = 0; sFlag[1] = 0;
            uint32_t* sDrop = reinterpret_cast<uint32_t*>(smem + A4::DROP_OFF);      // read by attn4_exact_tail
            sDrop[0] = p.drop_thr2; sDrop[1] = (uint32_t)p.drop_key; sDrop[2] = (uint32_t)(p.drop_key >> 32);
            sDrop[3] = __float_as_uint(((KMODE == 0 || DROP) && p.drop_on) ? p.drop_keep : 1.0f);
            sDrop[4] = (uint32_t)P;
            const uint64_t lp64 = reinterpret_cast<uint64_t>(p.lse);
            sDrop[5] = (uint32_t)lp64; sDrop[6] = (uint32_t)(lp64 >> 32);
            sDrop[7] = (uint32_t)p.side_post_drop;
        }
        if (((KMODE == 0 && p.chunk_id) || KMODE == 3) && tidb < LP) { sFirst[tidb] = LP; sLast[tidb] = -1; sCnt[tidb] = 0; }
        if (tidb < A4::NF) {
            const int j = tidb, jh = j / 192, jj = j % 192;
            sBias[j] = p.bqkv[(jj >> 6) * H + (a0 + jh) * 64 + (jj & 63)];
#ifdef MODCR_TUNING
            if constexpr (FOLD) reinterpret_cast<float*>(smem + A4::FOLD_OFF)[2 * LP + j] = p.cfold[(jj >> 6) * H + (a0 + jh) * 64 + (jj & 63)];
#endif
        }
#ifdef MODCR_TUNING
        if constexpr (FOLD) {
            if (tidb < LP) {
                float* sRs = reinterpret_cast<float*>(smem + A4::FOLD_OFF);
                const float2 st = tidb < L ? *reinterpret_cast<const float2*>(p.row_stats + ((int64_t)n * L + tidb) * 2) : float2{0.f, 0.f};
                sRs[tidb] = st.x; sRs[LP + tidb] = st.y;
            }
        }
#endif
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // half-tiles 0..5 = A0 B0 B1 A1 of K-tile 0, A0 B0 of K-tile 1
    stage_half(0, 0, 0); stage_half(0, 1, 0); stage_half(0, 2, 0); stage_half(0, 3, 0);
    stage_half(1, 0, 64); stage_half(1, 1, 64);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A4::FLY4) : "memory");      // A0, B0 of K-tile 0 landed
    __builtin_amdgcn_s_barrier();
    if ((wave >> 2) == 1) __builtin_amdgcn_s_barrier();     // group 1 (waves 4..7: one of each group per SIMD) runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);
    trace(1);
    for (int kt = 0; kt + 2 < nk; kt += 2) trip(std::integral_constant<int, 0>{}, kt);
    trip(std::integral_constant<int, 1>{}, nk - 2);
    if ((wave >> 2) == 0) __builtin_amdgcn_s_barrier();     // realign: every wave is done with the ring
    __builtin_amdgcn_sched_barrier(0);
    trace(2);
    // Per-tile copies of the lane indices the compiler cannot see through: everything below is loop-invariant
    // address arithmetic, and hoisted out of the tile loop it would pin > 100 registers across the K loop.
    int l15b = l15, l4b = l4, laneb = lane;
    asm volatile("" : "+v"(l15b), "+v"(l4b), "+v"(laneb));
    [[maybe_unused]] const int tid_d = wave * 64 + laneb;                // (per-tile opaque: the interleaved dump's addresses)

    // dense mask words of this wave's phase-B queries, issued now so they land under the image pass
    uint32_t wd[NQB][NKT];
    auto load_mask_words = [&]() {
        const int LWp = (L + 31) >> 5;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const int qi = (wave % A4::WPH) * QW + qb * 16 + l15b - P;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
                wd[qb][kt] = (qi >= 0 && qi < S && kt < LWp) ? p.bits[((int64_t)n * S + qi) * LWp + kt] : 0xffffffffu;
        }
    };
    if constexpr (KMODE == 2) load_mask_words();
    // ---- Q (scaled by log2(e)/8), K, V^T (+bias) as bf16 images: [Q0 | Q1 | K0 | K1 | Vt0 | Vt1] ------------
    const int hd_a = wc >> 1, part_a = wc & 1;              // this wave's q/k part and head in phase A
    {
        unsigned char* sQK = A4::img_qk(smem, part_a, hd_a);
        unsigned char* sVt = A4::img_vt(smem, hd_a);
        const float* bqk = sBias + hd_a * 192 + part_a * 64;
        const float* bv = sBias + hd_a * 192 + 128;
        const float qs = part_a == 0 ? 0.125f * LOG2E : 1.0f;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int tb = wr * A4::RW + mh * QW + i * 16;
                [[maybe_unused]] const float* sRs = reinterpret_cast<const float*>(smem + A4::FOLD_OFF);
                [[maybe_unused]] const float* sCf = sRs + 2 * LP + hd_a * 192;
                [[maybe_unused]] float rs_t = 1.f, rm_t = 0.f;          // FOLD: rstd and rstd * mu of this lane's q / k token
                if constexpr (FOLD) { rs_t = sRs[tb + l15b]; rm_t = sRs[LP + tb + l15b]; }
#pragma unroll
                for (int b = 0; b < 4; ++b) {               // q/k feature blocks: d0 = 16 b
                    const f32x4& v = acc[mh][b == 3][i][b == 3 ? 0 : b];
                    const int d0 = 16 * b + 4 * l4b;
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(bqk + d0);
                    bf16x4 o;
                    if constexpr (FOLD) {                   // LayerNorm folded into the projection: rstd acc - rstd mu c + d
                        const f32x4 cf = *reinterpret_cast<const f32x4*>(sCf + part_a * 64 + d0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)((fmaf(v[e], rs_t, bs[e]) - rm_t * cf[e]) * qs);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)((v[e] + bs[e]) * qs);
                    }
                    *reinterpret_cast<bf16x4*>(sQK + swz128(tb + l15b, d0 >> 3) + (d0 & 7) * 2) = o;
                }
#pragma unroll
                for (int j = 1; j < 3; ++j) {               // v feature blocks
                    const int d = part_a * 32 + (j - 1) * 16 + l15b;
                    const float bb = bv[d];
                    const f32x4& v = acc[mh][1][i][j];
                    bf16x4 o;
                    if constexpr (FOLD) {                   // four tokens of one feature per lane
                        const f32x4 r4 = *reinterpret_cast<const f32x4*>(sRs + tb + 4 * l4b), m4 = *reinterpret_cast<const f32x4*>(sRs + LP + tb + 4 * l4b);
                        const float cv = sCf[128 + d];
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)(fmaf(v[e], r4[e], bb) - m4[e] * cv);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)(v[e] + bb);
                    }
                    *reinterpret_cast<bf16x4*>(sVt + d * VT_STRIDE + (tb + 4 * l4b) * 2) = o;
                }
            }
    }
    __syncthreads();
    trace(3);

    const int hd = wave / A4::WPH, qbase = (wave % A4::WPH) * QW;      // phase B: head, first query
    unsigned char* sQ = A4::img_qk(smem, 0, hd);
    unsigned char* sK = A4::img_qk(smem, 1, hd);
    unsigned char* sVt = A4::img_vt(smem, hd);
    const int a = a0 + hd;

    // ---- chunk-mean query (phase-3 layers of seq_enc, v10:66-78): out of line, see attn4_chunk_mean ---------
    if constexpr (KMODE == 0 || KMODE == 3) {
        if ((KMODE == 3 || p.chunk_id) && !MODCR_DBG(p.debug & 16))        // debug bit 4: timing-only, no chunk means
            attn4_chunk_mean<LP, NHD>(smem, p.chunk_t, tid);
    }
    if (MODCR_DBG(p.debug & 1)) { __syncthreads(); continue; }
    // ---- training forward of a layer that will be differentiated: the images leave as rows (Q scaled by log2e / 8 and
    // chunk-averaged exactly as phase B sees it; V transposed back by ds_read_b64_tr_b16), 72 KB per head at LP = 192
    if constexpr (NHD == 2 && !DUMPV) {
        if (p.dump) {
            int td = tid;
            asm volatile("" : "+v"(td));
            typedef __attribute__((address_space(3))) bf16x4* lds_tr;
#pragma unroll 1
            for (int hh = 0; hh < 2; ++hh) {
                bf16* dst = p.dump + ((int64_t)n * p.A + a0 + hh) * (3 * LP * 64);
                const unsigned char* iq = A4::img_qk(smem, 0, hh);
                const unsigned char* ik = A4::img_qk(smem, 1, hh);
                const unsigned char* iv = A4::img_vt(smem, hh);
#pragma unroll 1
                for (int it = td; it < LP * 8; it += A4::NT) {
                    const int r = it >> 3, c = it & 7;
                    *reinterpret_cast<uint4*>(dst + r * 64 + c * 8) = *reinterpret_cast<const uint4*>(iq + swz128(r, c));
                    *reinterpret_cast<uint4*>(dst + LP * 64 + r * 64 + c * 8) = *reinterpret_cast<const uint4*>(ik + swz128(r, c));
                }
                // V rows: a 16-lane group reads the 4 x 16 block (features 8 c .. + 3 | + 4 .. + 7, keys 16 kb ..) of the V^T image
                // transposed: lane i receives key 16 kb + i, eight consecutive features = one 16-byte piece of its row
#pragma unroll 1
                for (int it = td; it < LP * 8; it += A4::NT) {          // LP * 8 / 512 whole trips: EXEC stays full
                    const int pi = it >> 4, i = it & 15, kb16 = pi >> 3, c = pi & 7;
                    const unsigned char* at = iv + (c * 8 + (i >> 2)) * VT_STRIDE + (kb16 * 16 + (i & 3) * 4) * 2;
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr)(at));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr)(at + 4 * VT_STRIDE));
                    bf16x8 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                    *reinterpret_cast<bf16x8*>(dst + 2 * LP * 64 + (kb16 * 16 + i) * 64 + c * 8) = v;
                }
            }
        }
    }
    if constexpr (KMODE == 3) load_mask_words();            // after the call above (18 registers it would have to save)

    // ---- phase B ----------------------------------------------------------------------------------------
    // S^T block (kt, kb, qb): key = 32 kt + 16 kb + 4 l4b + e in register e, query = qbase + 16 qb + l15b.
    if constexpr (KMODE == 0) {
        attn4_exact_tail<LP, NHD>(smem, p.bits, p.probs, p.align_map, p.ctx, p.align_t, S, H, p.A, n, a0, tid);
        break;                                              // the generic variant is launched one tile per workgroup
    } else {
        // streaming pass: P' = exp2(S) with no row max (scores are log2-domain, masked keys sit at -14427 or
        // -inf and flush to 0), tile kt+1's Q.K^T issued ahead of tile kt's exponentials; the row sum comes
        // out of the matrix pipe as a fifth V^T block of ones (summed over the bf16 P' the numerator uses).
        f32x4 o[4][NQB];
        float inv[NQB];
        bf16x8 fq[NQB][2];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                fq[qb][ks] = *reinterpret_cast<const bf16x8*>(sQ + swz128(qbase + qb * 16 + l15b, ks * 4 + l4b));
        bf16x8 ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
        [[maybe_unused]] uint32_t dbase[NQB];               // dropout bases of this lane's query rows and key group l4b (attn_common.h)
        [[maybe_unused]] uint64_t dkey = p.drop_key;        // kept in a register pair: the addend of every v_mad_u64_u32 below
        if constexpr (DROP) {
            asm volatile("" : "+v"(dkey));
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) dbase[qb] = attn_drop_base((uint32_t)((n * p.A + a) * 256 + qbase + qb * 16 + l15b - P), (uint32_t)l4b, dkey);
        }
        f32x4 ol[NQB];
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            ol[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto qk_block = [&](int kt, int kb, int qb, const bf16x8& fk0, const bf16x8& fk1, const f32x4& mk) {
            f32x4 c = mk;
            if constexpr (KMODE == 2 || KMODE == 3) {       // dense mask: bit (16 kb + 4 l4b + e) of this query's word
                const uint32_t w2 = wd[qb][kt] >> (4 * l4b);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!((w2 >> (kb * 16 + e)) & 1u)) c[e] += MODCR_NEG * LOG2E;
            }
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk0, fq[qb][0], c, 0, 0, 0);
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk1, fq[qb][1], c, 0, 0, 0);
        };
        auto qk_tile = [&](auto KT_, f32x4 (&s)[NQB][2]) {
            constexpr int kt = decltype(KT_)::value;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int krow = kt * 32 + kb * 16;
                const bf16x8 fk0 = *reinterpret_cast<const bf16x8*>(sK + swz128(krow + l15b, l4b));
                const bf16x8 fk1 = *reinterpret_cast<const bf16x8*>(sK + swz128(krow + l15b, 4 + l4b));
                const f32x4 mk = *reinterpret_cast<const f32x4*>(sMask + krow + 4 * l4b);
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) s[qb][kb] = qk_block(kt, kb, qb, fk0, fk1, mk);
            }
        };
        auto pv_tile = [&](auto KT_, const f32x4 (&s)[NQB][2]) {
            constexpr int kt = decltype(KT_)::value;
            // (round 6, measured and removed: alternating s_setprio between the two wave halves key tile by key tile, so that the younger
            // wave of a SIMD -- which finishes this VALU-bound phase alone, 13.2 k against 8.1 k cycles -- is not left with a solo tail:
            // 378.7 -> 377.8 us per tile, every second tile 381.5; profiles/r06_ab_attn_skip_and_alternate.log)
            bf16x8 va[4];
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const unsigned char* vrow = sVt + (db * 16 + l15b) * VT_STRIDE + (kt * 32 + 4 * l4b) * 2;
                const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vrow);
                const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vrow + 32);
                va[db][0] = lo[0]; va[db][1] = lo[1]; va[db][2] = lo[2]; va[db][3] = lo[3];
                va[db][4] = hi[0]; va[db][5] = hi[1]; va[db][6] = hi[2]; va[db][7] = hi[3];
            }
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                bf16x8 pb;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pb[4 * kb + e] = (bf16)__builtin_amdgcn_exp2f(s[qb][kb][e]);
                ol[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb, ol[qb], 0, 0, 0);     // row sum: unmasked weights
                if constexpr (DROP) pb = attn_drop8<kt>(pb, dbase[qb], dkey, p.drop_thr2);
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[db], pb, o[db][qb], 0, 0, 0);
            }
        };
        // interleaved dump (DUMPV): piece q of this thread = (head, Q | K | V, item td + 512 itn), three pieces per key tile
        [[maybe_unused]] auto dump_slot = [&](auto SLOT_) {
            if constexpr (DUMPV) {
                constexpr int slot = decltype(SLOT_)::value;
                constexpr int IT = LP * 8 / A4::NT;                     // items per thread, head and image (3 / 2)
                static_assert(2 * 3 * IT == 3 * NKT, "three dump pieces per key tile");
                typedef __attribute__((address_space(3))) bf16x4* lds_tr;
#pragma unroll
                for (int q = 3 * slot; q < 3 * slot + 3; ++q) {
                    const int hh = q / (3 * IT), kind = (q % (3 * IT)) / IT, it = tid_d + (q % IT) * A4::NT;
                    bf16* dst = p.dump + ((int64_t)n * p.A + a0 + hh) * (3 * LP * 64);
                    if (kind < 2) {
                        const int r = it >> 3, c = it & 7;
                        const unsigned char* img = A4::img_qk(smem, kind, hh);
                        *reinterpret_cast<uint4*>(dst + kind * LP * 64 + r * 64 + c * 8) = *reinterpret_cast<const uint4*>(img + swz128(r, c));
                    } else {
                        // V rows: a 16-lane group reads the 4 x 16 block of the V^T image transposed (see the block form above)
                        const unsigned char* iv = A4::img_vt(smem, hh);
                        const int pi = it >> 4, i = it & 15, kb16 = pi >> 3, c = pi & 7;
                        const unsigned char* at = iv + (c * 8 + (i >> 2)) * VT_STRIDE + (kb16 * 16 + (i & 3) * 4) * 2;
                        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr)(at));
                        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr)(at + 4 * VT_STRIDE));
                        bf16x8 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
                        *reinterpret_cast<bf16x8*>(dst + 2 * LP * 64 + (kb16 * 16 + i) * 64 + c * 8) = v;
                    }
                }
            }
        };
        {
            f32x4 sA[NQB][2], sB[NQB][2];
            qk_tile(std::integral_constant<int, 0>{}, sA);
            qk_tile(std::integral_constant<int, 1>{}, sB); dump_slot(std::integral_constant<int, 0>{}); pv_tile(std::integral_constant<int, 0>{}, sA);
            qk_tile(std::integral_constant<int, 2>{}, sA); dump_slot(std::integral_constant<int, 1>{}); pv_tile(std::integral_constant<int, 1>{}, sB);
            qk_tile(std::integral_constant<int, 3>{}, sB); dump_slot(std::integral_constant<int, 2>{}); pv_tile(std::integral_constant<int, 2>{}, sA);
            if constexpr (NKT == 8) {
                qk_tile(std::integral_constant<int, 4>{}, sA); pv_tile(std::integral_constant<int, 3>{}, sB);
                qk_tile(std::integral_constant<int, 5>{}, sB); pv_tile(std::integral_constant<int, 4>{}, sA);
                qk_tile(std::integral_constant<int, 6>{}, sA); pv_tile(std::integral_constant<int, 5>{}, sB);
                qk_tile(std::integral_constant<int, 7>{}, sB); pv_tile(std::integral_constant<int, 6>{}, sA);
                pv_tile(std::integral_constant<int, 7>{}, sB);
            } else if constexpr (NKT == 6) {
                qk_tile(std::integral_constant<int, 4>{}, sA); dump_slot(std::integral_constant<int, 3>{}); pv_tile(std::integral_constant<int, 3>{}, sB);
                qk_tile(std::integral_constant<int, 5>{}, sB); dump_slot(std::integral_constant<int, 4>{}); pv_tile(std::integral_constant<int, 4>{}, sA);
                dump_slot(std::integral_constant<int, 5>{}); pv_tile(std::integral_constant<int, 5>{}, sB);
            } else {
                dump_slot(std::integral_constant<int, 3>{}); pv_tile(std::integral_constant<int, 3>{}, sB);
            }
        }
        bool ok = true;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const float l = ol[qb][0];                      // every register of the ones block holds the row sum
            ok = ok && (l > 1e-30f) && (l < 1e30f);
            inv[qb] = 1.0f / l;
        }

        // a row sum out of range anywhere in the workgroup -> everybody redoes the tile with the exact pass
        if ((!__all(ok) || MODCR_DBG(p.debug & 8)) && laneb == 0) *sFlag = 1;
        trace(4);
        __syncthreads();
        trace(5);
        // The exact pass is called AFTER the common path's block, from its own re-read of the flag: inside an if / else
        // with the call in one arm, the compiler parks the accumulators in scratch ahead of the branch on every tile
        // (measured: 100 MB of scratch writes per launch).
        if (!*sFlag) {
            if (p.lse) {                                    // row statistics for the backward: log2 of the (unmasked) row sum
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) {
                    const int qi = qbase + qb * 16 + l15b - P;
                    if (l4b == 0 && qi >= 0 && qi < S) p.lse[((int64_t)n * p.A + a) * S + qi] = -__builtin_amdgcn_logf(inv[qb]);
                }
            }
            // context rows first (Q frags are in registers: the wave's own Q rows are free for the transpose), so that
            // the accumulators are dead during the align-map pass
            if constexpr (DROP) {                           // the context rows carry dropout's 1 / (1 - p)
                float inv_ctx[NQB];
#pragma unroll
                for (int qb = 0; qb < NQB; ++qb) inv_ctx[qb] = inv[qb] * p.drop_keep;
                attn4_store_ctx<A4>(o, inv_ctx, sQ + qbase * 128, p.ctx + ((int64_t)n * S + qbase - P) * H + a * 64, H, P - qbase, L - qbase, l15b, l4b, laneb);
            } else {
                attn4_store_ctx<A4>(o, inv, sQ + qbase * 128, p.ctx + ((int64_t)n * S + qbase - P) * H + a * 64, H, P - qbase, L - qbase, l15b, l4b, laneb);
            }
            if (KMODE == 3 && !MODCR_DBG(p.debug & 32)) {                // debug bit 5: timing-only, no align map
                // head-summed text -> region block, no LDS atomics: each head has its OWN [T][R] tile in LDS (head 0 over the
                // V^T images, head 1 over the Q images: both dead once every wave is past the barrier below; K stays, the
                // scores are recomputed from it) and a wave owns its 48 query rows, so the P' / l values go out as plain
                // 16-byte stores straight from the recomputed score blocks.  The sum of the two tiles then goes to the [N,T,R]
                // map in HBM as float atomics (six head-pair tiles and three layers meet there; fire-and-forget: a
                // read-add-write by an owning workgroup measured slower).
                const int T = p.align_t, R = S - T;
                // (one head per workgroup, LP = 256: its tile over the V^T image; the launcher checks that it fits)
                float* sAm = reinterpret_cast<float*>(hd == 0 ? A4::img_vt(smem, 0) : smem);
                const bool vec = ((T | R) & 3) == 0;            // 16-byte pieces: a lane's four keys are in or out together
                __syncthreads();                                    // every wave is done with Q rows (context transposes) and V^T
                if (qbase < T) {
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt)
                        if (kt * 32 + 31 >= T) {
#pragma unroll
                            for (int kb = 0; kb < 2; ++kb) {
                                const int krow = kt * 32 + kb * 16, key0 = krow + 4 * l4b;
                                const bf16x8 fk0 = *reinterpret_cast<const bf16x8*>(sK + swz128(krow + l15b, l4b));
                                const bf16x8 fk1 = *reinterpret_cast<const bf16x8*>(sK + swz128(krow + l15b, 4 + l4b));
                                const f32x4 mk = *reinterpret_cast<const f32x4*>(sMask + krow + 4 * l4b);
#pragma unroll
                                for (int qb = 0; qb < NQB; ++qb) {
                                    const int qi = qbase + qb * 16 + l15b;
                                    const f32x4 sv = qk_block(kt, kb, qb, fk0, fk1, mk);
                                    f32x4 v;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_exp2f(sv[e]) * inv[qb];
                                    if constexpr (DROP) {
                                        if (p.side_post_drop) {         // the map sums the probabilities AFTER the dropout (v10:94-106)
                                            const int thr_s = (int)(short)(p.drop_thr2 & 0xffffu) + 1;
#pragma unroll
                                            for (int i = 0; i < 2; ++i) {
                                                const uint32_t w = attn_drop_word(dbase[qb], (2 * kt + kb) * 2 + i, dkey);
#pragma unroll
                                                for (int f = 0; f < 2; ++f)
                                                    v[2 * i + f] = (int)(short)(w >> (16 * f)) >= thr_s ? v[2 * i + f] * p.drop_keep : 0.f;
                                            }
                                        }
                                    }
                                    if (qi >= T) continue;
                                    float* at = sAm + qi * R + (key0 - T);
                                    if (vec) {
                                        if (key0 >= T && key0 < L) *reinterpret_cast<f32x4*>(at) = v;
                                    } else {
#pragma unroll
                                        for (int e = 0; e < 4; ++e)
                                            if (key0 + e >= T && key0 + e < L) at[e] = v[e];
                                    }
                                }
                            }
                        }
                }
                __syncthreads();
                const float* sA0 = reinterpret_cast<const float*>(A4::img_vt(smem, 0));
                const float* sA1 = reinterpret_cast<const float*>(smem);
                float* dst = p.align_map + (int64_t)n * T * R;
                // (debug bits 6 / 7, timing only: no output at all / plain stores instead of atomics -- what VERDICT r04 item 2b's
                // per-head-pair partial tiles would issue, without their fold: tools/ab_align_atomics.py)
                if (MODCR_DBG(p.debug & 64)) {
                } else if (MODCR_DBG(p.debug & 128)) {
                    for (int j = tid; j < T * R; j += A4::NT) dst[j] = sA0[j] + (NHD == 2 ? sA1[j] : 0.f);
                } else if constexpr (NHD == 2) {
                    for (int j = tid; j < T * R; j += A4::NT) atomicAdd(dst + j, sA0[j] + sA1[j]);
                } else {
                    for (int j = tid; j < T * R; j += A4::NT) atomicAdd(dst + j, sA0[j]);
                }
            }
        }
        asm volatile("" ::: "memory");
        if (*reinterpret_cast<volatile int*>(sFlag))
            attn4_exact_tail<LP, NHD>(smem, p.bits, nullptr, KMODE == 3 ? p.align_map : nullptr, p.ctx, p.align_t, S, H, p.A, n, a0, tid);
    }
    trace(6);
    __syncthreads();        // the images and tables are dead: the next tile's tables / prologue may overwrite them
    trace(7);
    ++trace_it;
    }   // tiles
}

template <int MODE, int LP, int DROP, int NH = 2, int DUMPV = 0, int FOLD = 0>
int launch_attn4d(const AttnArgs& p, hipStream_t st) {
    typedef A4T<LP, NH> A4;
#ifndef MODCR_TUNING
    static_assert(FOLD == 0, "the LayerNorm-fold prototype exists in the tuning library only");
#endif
    constexpr int SMEM_ = FOLD ? A4::SMEM_FOLD : A4::SMEM;
    if constexpr (NH == 2 && MODE != 0 && LP != 256 && DUMPV == 0) {
        if (p.dump && !modcr_knob_set("MODCR_ATTN_DUMP_BLOCK")) return launch_attn4d<MODE, LP, DROP, NH, 1>(p, st);      // (knob: tuning build, A/B)
    }
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_attn4_kernel<MODE, LP, DROP, NH, DUMPV A4_FOLD_ARG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_);
        if (e != hipSuccess) {
            modcr_set_error("qkv_attn4: cannot reserve %d bytes of LDS: %s", SMEM_, hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    const int ncu = modcr_device_cus() >= 8 ? (modcr_device_cus() & ~7) : 256;
    const int nopersist = modcr_knob_set("MODCR_ATTN_NOPERSIST");          // tuning build only
    const int ntiles = p.N * (p.A / NH);
    const int grid = (ntiles <= ncu || nopersist || MODE == 0) ? ntiles : ncu;
    launch_timed(qkv_attn4_kernel<MODE, LP, DROP, NH, DUMPV A4_FOLD_ARG>, dim3(grid), dim3(A4::NT), (size_t)SMEM_, st, p);
    return modcr_check_launch("qkv_attn4");
}
template <int MODE, int LP, int NH = 2>
int launch_attn4(const AttnArgs& p, hipStream_t st) {
    // the streaming variants carry the dropout masking as a template flag (eval-mode code unchanged); the generic
    // variant (MODE 0) always ends in the exact pass, which takes the threshold at run time
    if (MODE != 0 && p.drop_on) return launch_attn4d<MODE, LP, (MODE != 0), NH>(p, st);
    return launch_attn4d<MODE, LP, 0, NH>(p, st);
}

#ifdef MODCR_TUNING
// LayerNorm-fold PROTOTYPE (tuning library only; tools/proto_ln_fold.py): the key-mask call of the 192-token tile on PRE-LayerNorm rows.
//   pre [N,S,H] bf16, wfold [3H,H] = bf16(gamma o Wqkv), d [3H] = Wqkv beta + bqkv, c [3H] = row sums of wfold (as the MFMA sees them),
//   row_stats [N*S][2] = (rstd, rstd * mu) of every row of pre
extern "C" int modcr_tuning_qkv_attn_fold_fwd(const void* pre, const void* wfold, const float* d, const float* c, const float* row_stats,
                                              const float* key_mask, void* ctx, int32_t N, int32_t S, int32_t H, int32_t A, float attn_p,
                                              uint64_t seed, uint64_t offset, modcr_stream_t stream) {
    MODCR_REQUIRE(pre && wfold && d && c && row_stats && key_mask && ctx, "qkv_attn_fold_fwd: null pointer");
    MODCR_REQUIRE(S > 128 && S <= 192 && (A % 2) == 0 && H == A * 64 && (H % 128) == 0, "qkv_attn_fold_fwd: the 192-token tile only");
    AttnArgs p;
    p.x = (const bf16*)pre; p.hist = nullptr; p.wqkv = (const bf16*)wfold; p.bqkv = d; p.key_mask = key_mask; p.bits = nullptr; p.chunk_id = nullptr;
    p.ctx = (bf16*)ctx; p.probs = nullptr; p.align_map = nullptr; p.lse = nullptr; p.dump = nullptr;
    p.N = N; p.S = S; p.P = 0; p.H = H; p.A = A; p.chunk_t = 0; p.align_t = 0; p.hconc = 3; p.debug = modcr_knob_int("MODCR_ATTN_DEBUG", 0);
    p.drop_thr2 = 0; p.drop_on = 0; p.drop_key = 0; p.drop_keep = 1.f; p.side_post_drop = 0;
    if (attn_p > 0.f) {
        p.drop_key = seed + offset * 0x9E3779B97F4A7C15ull; p.drop_thr2 = attn_thrm1_2(attn_thr16(attn_p)); p.drop_on = 1; p.drop_keep = 1.0f / (1.0f - attn_p);
    }
    p.row_stats = row_stats; p.cfold = c;
    return attn_p > 0.f ? launch_attn4d<1, 192, 1, 2, 0, 1>(p, (hipStream_t)stream) : launch_attn4d<1, 192, 0, 2, 0, 1>(p, (hipStream_t)stream);
}
#endif

// [prefix ; x] rows of every sequence as one buffer (the tile kernels stage their token rows from ONE base + 32-bit offsets):
// out [N, P + S, H] <- hist [N, P, H], x [N, S, H], 16-byte pieces
__global__ __launch_bounds__(256) void concat_prefix_kernel(const bf16* hist, const bf16* x, bf16* out, int N, int P, int S, int H) {
    const int hv = H >> 3, L = P + S;
    const int64_t total = (int64_t)N * L * hv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % hv);
        const int64_t row = i / hv;
        const int r = (int)(row % L);
        const int64_t n = row / L;
        const bf16* src = r < P ? hist + (n * P + r) * H : x + (n * S + (r - P)) * H;
        reinterpret_cast<uint4*>(out)[i] = reinterpret_cast<const uint4*>(src)[c];
    }
}

// ---- fp32 parity core: one block per (n, head); K_h and V_h in LDS, one query per wave-iteration
struct AttnF32Args {
    const float* qkv_x;   // [N,S,3H]
    const float* qkv_h;   // [N,P,3H] or null
    const float* key_mask; const uint32_t* bits;
    float* ctx; float* probs; float* align_map;
    int N, S, P, H, A, align_t;
    // attention-probability dropout (attn_common.h; the same mask as the bf16 kernels): 0 = off, else round(p * 2^16); side outputs after it?
    uint32_t drop_thr16; uint64_t drop_key; float drop_keep; int side_post_drop;
};

__global__ __launch_bounds__(256) void attn_f32_kernel(AttnF32Args p) {
    extern __shared__ float sm[];
    const int S = p.S, P = p.P, L = P + S, H = p.H;
    float* sK = sm;                  // L x 65
    float* sV = sK + (size_t)L * 65; // L x 65
    float* sQ = sV + (size_t)L * 65; // 4 x 64
    float* sP = sQ + 256;            // 4 x L
    const int n = blockIdx.x / p.A, a = blockIdx.x % p.A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int idx = tid; idx < L * 64; idx += 256) {
        const int j = idx >> 6, d = idx & 63;
        const float* row = (j < P) ? p.qkv_h + ((int64_t)n * P + j) * 3 * H
                                   : p.qkv_x + ((int64_t)n * S + (j - P)) * 3 * H;
        sK[j * 65 + d] = row[H + a * 64 + d];
        sV[j * 65 + d] = row[2 * H + a * 64 + d];
    }
    __syncthreads();
    const int LW = (L + 31) >> 5;
    for (int i = wave; i < S; i += 4) {
        sQ[wave * 64 + lane] = p.qkv_x[((int64_t)n * S + i) * 3 * H + a * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        float sv[4];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = lane + 64 * c;
            sv[c] = -INFINITY;
            if (j < L) {
                float s = 0.f;
                for (int d = 0; d < 64; ++d) s = fmaf(sQ[wave * 64 + d], sK[j * 65 + d], s);
                s = s / 8.0f;
                bool see;
                if (p.bits) see = (p.bits[((int64_t)n * S + i) * LW + (j >> 5)] >> (j & 31)) & 1u;
                else see = p.key_mask[(int64_t)n * L + j] != 0.f;
                s += see ? 0.f : MODCR_NEG;
                sv[c] = s;
                mx = fmaxf(mx, s);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = lane + 64 * c;
            if (j < L) { sv[c] = expf(sv[c] - mx); sum += sv[c]; }
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = lane + 64 * c;
            if (j < L) {
                const float pj = sv[c] / sum;
                float pd = pj;                              // after nn.Dropout (modeling_bert.py:69): what multiplies V
                if (p.drop_thr16) {
                    uint32_t hx, hy;
                    attn_drop_words((uint32_t)((n * p.A + a) * 256 + i), j >> 2, p.drop_key, hx, hy);
                    pd = attn_keep_field(hx, hy, j & 3, p.drop_thr16) ? pj * p.drop_keep : 0.f;
                }
                sP[wave * L + j] = pd;
                const float side = p.side_post_drop ? pd : pj;
                if (p.probs) p.probs[(((int64_t)n * p.A + a) * S + i) * L + j] = side;
                if (p.align_map && i < p.align_t && j >= P + p.align_t)
                    atomicAdd(p.align_map + ((int64_t)n * p.align_t + i) * (S - p.align_t) + (j - P - p.align_t), side);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        float o = 0.f;
        for (int j = 0; j < L; ++j) o = fmaf(sP[wave * L + j], sV[j * 65 + lane], o);
        p.ctx[((int64_t)n * S + i) * H + a * 64 + lane] = o;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
}



// ---- attention core backward on the matrix pipe (bf16 path, S <= 192): one workgroup of 6 waves per (n, head) -------
// Same inputs / outputs as attn_bwd_f32_kernel below (recomputed fp32 q|k|v rows, dctx -> fp32 dq|dk|dv rows).
// Wave w owns tokens 32 w .. 32 w + 31 twice:
//   sub-pass Q (its QUERIES; K rows, V rows and K^T in LDS): S^T = K.Qs^T for all 192 keys stays in registers
//     (Qs = Q log2e / 8, mask = accumulator init) -> exact row max / sum -> P^T; delta = rowsum(P o dP) with
//     dP^T = V.dO^T; then dS^T = P^T o (dP^T - delta), whose bf16 pairs are the B operand of dQ^T += K^T.dS^T
//     (accumulator as operand, keys permuted identically on the K^T fragment reads).  Row statistics go to LDS.
//   sub-pass K (its KEYS; Qs rows, dO rows, Qs^T, dO^T in LDS): per 32-query tile S = Qs.K^T and dP = dO.V^T with
//     the query in the registers, P and dS rebuilt from the row statistics, dV^T += dO^T.P, dK^T += Qs^T.dS.
// Padding: dO rows beyond S are zero (their P, dS contribute nothing), keys beyond S carry a -inf mask.
struct AB {
    static constexpr int LP = 192, NT = 384;
    static constexpr int ROWS = LP * 128;                   // [token][64] bf16, 128-byte rows, swz128
    // The transposed operands (K^T; Qs^T, dO^T) are NOT kept as images: ds_read_b64_tr_b16 reads them out of the row images
    // (a lane receives four consecutive tokens of one feature = half an MFMA operand).  Round 1 built [d][token] copies with
    // eight 2-byte LDS stores per 16-byte piece and needed 108 KB per workgroup (one workgroup of six waves per CU);
    // 57 KB now, two workgroups per CU.
    static constexpr int IMG = 2 * ROWS;                    // sub-pass Q: K | V    sub-pass K: Qs | dO
    static constexpr int SMEM = IMG + 4 * LP * 4 + LP * 6 * 4;
};

// DALIGN: the align map's gradient rides in (seq_enc layers 9-11 of the trainable-encoder variant only).  A template parameter:
// as a run-time branch its per-block address arithmetic was hoisted to the top of the kernel and spilled (85 scratch stores
// before the first MFMA) in every call.
template <typename TD, bool DALIGN>
__global__ __launch_bounds__(384, 3) void attn_bwd_mfma_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LP = AB::LP;
    unsigned char* img0 = smem;                             // rows image 0 (K | Qs)
    unsigned char* img1 = smem + AB::ROWS;                  // rows image 1 (V | dO)
    float* sM = reinterpret_cast<float*>(smem + AB::IMG);
    float* sInv = sM + LP;
    float* sDl = sInv + LP;
    float* sMask = sDl + LP;
    uint32_t* sBits = reinterpret_cast<uint32_t*>(sMask + LP);
    const int S = p.S, H = p.H;
    const int n = blockIdx.x / p.A, a = blockIdx.x % p.A;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15_ = lane & 15, l4_ = lane >> 4;
    const int LW = (S + 31) >> 5;
    const bf16* qkv = p.qkvb + (int64_t)n * S * 3 * H + a * 64;
    const TD* dctx = reinterpret_cast<const TD*>(p.dctx) + (int64_t)n * S * H + a * 64;
    float* dqkv = p.dqkv + (int64_t)n * S * 3 * H + a * 64;
    constexpr float QS = 0.125f * LOG2E;

    // 8 consecutive features of token t (zeros beyond S) from the bf16 q|k|v rows (k, v: as they are -- the values the forward's
    // images held; q: scaled by log2e / 8 and rounded again) or from dctx
    auto row8 = [&](int t, int part, int c, float scale) {
        bf16x8 o;
        if (t < S) {
            o = *reinterpret_cast<const bf16x8*>(qkv + (int64_t)t * 3 * H + part * H + c * 8);
            if (scale != 1.0f) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)o[e] * scale);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
        }
        return o;
    };
    auto do8 = [&](int t, int c) {
        bf16x8 o;
        if (t < S) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)to_f32(dctx[(int64_t)t * H + c * 8 + e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)0.f;
        }
        return o;
    };
    auto put_rows = [&](unsigned char* img, int t, int c, const bf16x8& v) { *reinterpret_cast<bf16x8*>(img + swz128(t, c)) = v; };
    // Transposed fragment of a row image: feature 16 db + l15, tokens tok0 + 4 l4 + {0..3} and tok0 + 16 + 4 l4 + {0..3} (the key
    // order of the accumulator-as-operand products).  Within a 16-lane group lane i addresses token row (i >> 2), 8-byte piece
    // (i & 3) of the 16-feature span and receives feature i (tools: transpose64_kernel, the TN GEMM); swz128 is applied per lane.
    typedef __attribute__((address_space(3))) bf16x4* lds_tr;
    auto tr8 = [&](const unsigned char* img, int tok0, int db, int l15, int l4) {
        const int r = tok0 + 4 * l4 + (l15 >> 2);
        const int ch = db * 2 + ((l15 & 3) >> 1), within = (l15 & 1) * 8;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr)(img + swz128(r, ch) + within));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr)(img + swz128(r + 16, ch) + within));
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
        return o;
    };

    // ---- tables and the images of sub-pass Q ---------------------------------------------------------------
    for (int j = tid; j < LP; j += AB::NT) {
        float m;
        if (j >= S) m = -INFINITY;
        else if (p.bits) m = 0.f;
        else m = (1.0f - p.key_mask[(int64_t)n * S + j]) * (MODCR_NEG * LOG2E);
        sMask[j] = m;
    }
    if (p.bits)
        for (int j = tid; j < LP * 6; j += AB::NT) {
            const int q = j / 6, w = j % 6;
            sBits[j] = (q < S && w < LW) ? p.bits[((int64_t)n * S + q) * LW + w] : 0xffffffffu;
        }
    for (int it = tid; it < LP * 8; it += AB::NT) {
        const int t = it >> 3, c = it & 7;
        put_rows(img0, t, c, row8(t, 1, c, 1.0f));
        put_rows(img1, t, c, row8(t, 2, c, 1.0f));
    }
    __syncthreads();

    const int t0 = wave * 32;                               // this wave's tokens
    if (MODCR_DBG(p.debug & 1)) return;
    // ---- sub-pass Q ------------------------------------------------------------------------------------------
    // One 16-query block at a time (its 16 x 192 scores = 48 registers): with both blocks of the wave's 32 queries unrolled
    // side by side the kernel took 256 registers + 42 spilled, one workgroup per CU; the K / V / K^T fragments are read twice.
    if (!MODCR_DBG(p.debug & 2)) {
#pragma unroll 1
        for (int qb = 0; qb < 2; ++qb) {
            // per-iteration opaque copies of the lane indices: the fragment addresses below are loop invariant, and hoisted out
            // of this loop their ~100 registers were spilled ahead of the first MFMA
            int l15 = l15_, l4 = l4_;
            asm volatile("" : "+v"(l15), "+v"(l4));
            const int qrow = t0 + qb * 16 + l15;
            bf16x8 fq[2], fdo[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fq[ks] = row8(qrow, 0, ks * 4 + l4, QS);            // zeros beyond S, as the Qs image of sub-pass K (same row statistics)
                fdo[ks] = do8(qrow, ks * 4 + l4);
            }
            f32x4 sc[6][2];
#pragma unroll
            for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const int krow = kt * 32 + kb * 16;
                    const bf16x8 fk0 = *reinterpret_cast<const bf16x8*>(img0 + swz128(krow + l15, l4));
                    const bf16x8 fk1 = *reinterpret_cast<const bf16x8*>(img0 + swz128(krow + l15, 4 + l4));
                    f32x4 c = *reinterpret_cast<const f32x4*>(sMask + krow + 4 * l4);
                    if (p.bits) {
                        const uint32_t word = sBits[qrow * 6 + kt];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (!((word >> (kb * 16 + 4 * l4 + e)) & 1u)) c[e] += MODCR_NEG * LOG2E;
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk0, fq[0], c, 0, 0, 0);
                    sc[kt][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk1, fq[1], c, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);      // no hoisting of the next blocks' fragment reads: registers
                }
            {
                float m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) m = fmaxf(m, sc[kt][kb][e]);
                m = fmaxf(m, __shfl_xor(m, 16, 64));
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float l = 0.f;
#pragma unroll
                for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float ex = __builtin_amdgcn_exp2f(sc[kt][kb][e] - m); sc[kt][kb][e] = ex; l += ex; }
                l += __shfl_xor(l, 16, 64);
                l += __shfl_xor(l, 32, 64);
                const float inv = 1.0f / l;
#pragma unroll
                for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) sc[kt][kb][e] *= inv;
                if (l4 == 0) { sM[qrow] = m; sInv[qrow] = inv; }
            }
            // forward dropout on the probabilities: O = (P o m) V with m = keep / (1 - p), so dP = m o (dO V^T) below
            const uint32_t dthr = p.drop_thr16;
            const uint32_t dbase = attn_drop_base((uint32_t)((n * p.A + a) * 256 + qrow), (uint32_t)l4, p.drop_key);
            auto dp_masked = [&](int kt, int kb) {
                const int krow = kt * 32 + kb * 16;
                const bf16x8 fv0 = *reinterpret_cast<const bf16x8*>(img1 + swz128(krow + l15, l4));
                const bf16x8 fv1 = *reinterpret_cast<const bf16x8*>(img1 + swz128(krow + l15, 4 + l4));
                f32x4 dp = {0.f, 0.f, 0.f, 0.f};
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv0, fdo[0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv1, fdo[1], dp, 0, 0, 0);
                float mk4[4] = {1.f, 1.f, 1.f, 1.f};
                if (dthr) {
                    uint32_t hx, hy;
                    hx = attn_drop_word(dbase, (2 * kt + kb) * 2, p.drop_key);
                    hy = attn_drop_word(dbase, (2 * kt + kb) * 2 + 1, p.drop_key);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { mk4[e] = attn_keep_field(hx, hy, e, dthr) ? p.drop_keep : 0.f; dp[e] *= mk4[e]; }
                }
                if constexpr (DALIGN) {         // the align map's gradient, text query x region key: on the un-dropped probabilities,
                                                // or (side_post_drop: the map summed P o m / (1 - p)) under the same mask
                    const int T = p.align_t, key0 = kt * 32 + kb * 16 + 4 * l4;
                    if (qrow < T && key0 + 3 >= T) {
                        const float* da = p.d_align + ((int64_t)n * T + qrow) * (S - T) - T;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (key0 + e >= T && key0 + e < S) dp[e] += p.side_post_drop ? mk4[e] * da[key0 + e] : da[key0 + e];
                    }
                }
                return dp;
            };
            float dl = 0.f;
#pragma unroll
            for (int kt = 0; kt < 6; ++kt)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const f32x4 dp = dp_masked(kt, kb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dl += sc[kt][kb][e] * dp[e];
                    __builtin_amdgcn_sched_barrier(0);
                }
            dl += __shfl_xor(dl, 16, 64);
            dl += __shfl_xor(dl, 32, 64);
            if (l4 == 0) sDl[qrow] = dl;
            f32x4 dq[4];
#pragma unroll
            for (int db = 0; db < 4; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < 6; ++kt) {
                bf16x8 dsb;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const f32x4 dp = dp_masked(kt, kb);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dsb[4 * kb + e] = (bf16)(sc[kt][kb][e] * (dp[e] - dl));
                }
#pragma unroll
                for (int db = 0; db < 4; ++db)
                    dq[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr8(img0, kt * 32, db, l15, l4), dsb, dq[db], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (qrow < S)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    f32x4 o = dq[db];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] *= 0.125f;
                    const int64_t at = (int64_t)qrow * 3 * H + db * 16 + 4 * l4;
                    if (p.out_bf16) *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16*>(p.dqkv) + (int64_t)n * S * 3 * H + a * 64 + at) = bf16x4{(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                    else *reinterpret_cast<f32x4*>(dqkv + at) = o;
                }
        }
    }
    __syncthreads();
    // ---- images of sub-pass K: Qs rows, dO rows, Qs^T, dO^T ------------------------------------------------------
    for (int it = tid; it < LP * 8; it += AB::NT) {
        const int t = it >> 3, c = it & 7;
        put_rows(img0, t, c, row8(t, 0, c, QS));
        put_rows(img1, t, c, do8(t, c));
    }
    __syncthreads();
    // ---- sub-pass K ------------------------------------------------------------------------------------------
    // (one 16-key block at a time, as sub-pass Q: the Qs / dO fragments are read twice)
    if (!MODCR_DBG(p.debug & 4)) {
#pragma unroll 1
        for (int kb = 0; kb < 2; ++kb) {
            const int key = t0 + kb * 16 + l15_;
            bf16x8 fkk[2], fvv[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fkk[ks] = row8(key, 1, ks * 4 + l4_, 1.0f);
                fvv[ks] = row8(key, 2, ks * 4 + l4_, 1.0f);
            }
            const float mkey = sMask[key];
            f32x4 dk[4], dv[4];
#pragma unroll
            for (int db = 0; db < 4; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
            for (int qt = 0; qt < 6; ++qt) {
                int l15 = l15_, l4 = l4_;
                asm volatile("" : "+v"(l15), "+v"(l4));
                bf16x8 pB, dsB;
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int qrow = qt * 32 + qb * 16;
                    const bf16x8 fq0 = *reinterpret_cast<const bf16x8*>(img0 + swz128(qrow + l15, l4));
                    const bf16x8 fq1 = *reinterpret_cast<const bf16x8*>(img0 + swz128(qrow + l15, 4 + l4));
                    const bf16x8 fd0 = *reinterpret_cast<const bf16x8*>(img1 + swz128(qrow + l15, l4));
                    const bf16x8 fd1 = *reinterpret_cast<const bf16x8*>(img1 + swz128(qrow + l15, 4 + l4));
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(sM + qrow + 4 * l4);
                    const f32x4 i4 = *reinterpret_cast<const f32x4*>(sInv + qrow + 4 * l4);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDl + qrow + 4 * l4);
                    f32x4 c = {mkey, mkey, mkey, mkey};
                    if (p.bits) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t word = sBits[(qrow + 4 * l4 + e) * 6 + (key >> 5)];
                            if (!((word >> (key & 31)) & 1u)) c[e] += MODCR_NEG * LOG2E;
                        }
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq0, fkk[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq1, fkk[1], c, 0, 0, 0);
                    f32x4 dp = {0.f, 0.f, 0.f, 0.f};
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd0, fvv[0], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd1, fvv[1], dp, 0, 0, 0);
                    // dropout: a lane's four values are four queries of ONE key, i.e. four different hash counters; the
                    // four lanes of a quad (keys of one key group) need the same four, so each hashes one and they are
                    // exchanged by quad broadcasts: one hash per lane instead of four
                    uint32_t hqx = 0, hqy = 0;
                    if (p.drop_thr16)
                        attn_drop_words((uint32_t)((n * p.A + a) * 256 + qrow + 4 * l4 + (l15 & 3)), key >> 2, p.drop_key, hqx, hqy);
                    const uint32_t xq[4] = {quad_bcast<0>(hqx), quad_bcast<1>(hqx), quad_bcast<2>(hqx), quad_bcast<3>(hqx)};
                    const uint32_t yq[4] = {quad_bcast<0>(hqy), quad_bcast<1>(hqy), quad_bcast<2>(hqy), quad_bcast<3>(hqy)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(c[e] - m4[e]) * i4[e];
                        float mk = 1.0f;                    // dropout factor of (query qrow + 4 l4 + e, key)
                        if (p.drop_thr16)                   // the words of query e were hashed by quad lane e
                            mk = attn_keep_field(xq[e], yq[e], key & 3, p.drop_thr16) ? p.drop_keep : 0.f;
                        float dpe = mk * dp[e];
                        if constexpr (DALIGN) {
                            const int T = p.align_t, qi = qrow + 4 * l4 + e;
                            if (qi < T && key >= T && key < S) {
                                const float da = p.d_align[((int64_t)n * T + qi) * (S - T) + (key - T)];
                                dpe += p.side_post_drop ? mk * da : da;
                            }
                        }
                        pB[4 * qb + e] = (bf16)(pe * mk);                   // dV takes the masked probabilities
                        dsB[4 * qb + e] = (bf16)(pe * (dpe - d4[e]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const bf16x8 qf = tr8(img0, qt * 32, db, l15, l4), df = tr8(img1, qt * 32, db, l15, l4);
                    dv[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, pB, dv[db], 0, 0, 0);
                    dk[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, dsB, dk[db], 0, 0, 0);
                }
            }
            if (key < S)
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    f32x4 ok = dk[db];
#pragma unroll
                    for (int e = 0; e < 4; ++e) ok[e] *= (1.0f / LOG2E);
                    const int64_t at = (int64_t)key * 3 * H + H + db * 16 + 4 * l4_;
                    if (p.out_bf16) {
                        bf16* ob = reinterpret_cast<bf16*>(p.dqkv) + (int64_t)n * S * 3 * H + a * 64 + at;
                        const f32x4 ov = dv[db];
                        *reinterpret_cast<bf16x4*>(ob) = bf16x4{(bf16)ok[0], (bf16)ok[1], (bf16)ok[2], (bf16)ok[3]};
                        *reinterpret_cast<bf16x4*>(ob + H) = bf16x4{(bf16)ov[0], (bf16)ov[1], (bf16)ov[2], (bf16)ov[3]};
                    } else {
                        *reinterpret_cast<f32x4*>(dqkv + at) = ok;
                        *reinterpret_cast<f32x4*>(dqkv + at + H) = dv[db];
                    }
                }
        }
    }
}

// ---- attention core backward, exact fp32 (VALU): one block (4 waves) per (n, head) ------------------------------
// From the recomputed q|k|v rows [N,S,3H] (fp32) and dctx: dq|dk|dv rows [N,S,3H] (fp32).  No saved
// probabilities: pass A (K, V in LDS; one query per wave-iteration) recomputes p_ij, forms dp_ij = dO_i.V_j,
// delta_i = sum_j p dp, ds_ij = p (dp - delta) and dQ_i = sum_j ds_ij K_j / 8, keeping (max, 1/sum, delta) of
// every row in LDS; pass B (Q, dO in LDS; one key per wave-iteration) rebuilds p and ds from those row
// statistics and accumulates dV_j = sum_i p_ij dO_i, dK_j = sum_i ds_ij Q_i / 8.  Mask semantics as the forward.

template <typename TD>
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(AttnBwdArgs p) {
    extern __shared__ float sm[];
    const int S = p.S, H = p.H;
    float* sA = sm;                          // pass A: K   | pass B: Q      [S][65]
    float* sB = sA + (size_t)S * 65;         // pass A: V   | pass B: dO     [S][65]
    float* sMx = sB + (size_t)S * 65;        // row max, 1 / row sum, delta  [3][S]
    float* sInv = sMx + S;
    float* sDl = sInv + S;
    float* sW = sDl + S;                     // per wave: 64 + 64 + S + S floats
    const int n = blockIdx.x / p.A, a = blockIdx.x % p.A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wq = sW + (size_t)wave * (128 + 2 * S);      // query row (A) / key row (B)
    float* wd = wq + 64;                                 // dO row (A) / value row (B)
    float* wp = wd + 64;                                 // p_j (B: p_i)
    float* ws = wp + S;                                  // ds_j (B: ds_i)
    const int LW = (S + 31) >> 5;
    const float* qkv = p.qkv + (int64_t)n * S * 3 * H;
    const TD* dctx = reinterpret_cast<const TD*>(p.dctx) + (int64_t)n * S * H;
    float* dqkv = p.dqkv + (int64_t)n * S * 3 * H;
    auto seen = [&](int i, int j) {
        if (p.bits) return ((p.bits[((int64_t)n * S + i) * LW + (j >> 5)] >> (j & 31)) & 1u) != 0;
        return p.key_mask[(int64_t)n * S + j] != 0.f;
    };
    auto dalign = [&](int i, int j) {       // gradient arriving through the align map
        const int T = p.align_t;
        return (p.d_align && i < T && j >= T) ? p.d_align[((int64_t)n * T + i) * (S - T) + (j - T)] : 0.f;
    };
    // the forward's attention-probability dropout (tile kernels, csrc/attn_common.h): keep / (1 - p) of weight (i, j), regenerated
    // from its counter layout -- ctx = (P o m) V, so dP = m o (dO V^T) and dV = (P o m)^T dO; the align map saw the unmasked P
    const uint32_t dthr = p.drop_thr16;
    auto dmask = [&](int i, int j) {
        if (!dthr) return 1.0f;
        uint32_t hx, hy;
        attn_drop_words((uint32_t)((n * p.A + a) * 256 + i), j >> 2, p.drop_key, hx, hy);
        return attn_keep_field(hx, hy, j & 3, dthr) ? p.drop_keep : 0.f;
    };
    // ---- pass A ------------------------------------------------------------------------------------------
    for (int idx = tid; idx < S * 64; idx += 256) {
        const int j = idx >> 6, d = idx & 63;
        sA[j * 65 + d] = qkv[(int64_t)j * 3 * H + H + a * 64 + d];
        sB[j * 65 + d] = qkv[(int64_t)j * 3 * H + 2 * H + a * 64 + d];
    }
    __syncthreads();
    for (int i = wave; i < S; i += 4) {
        wq[lane] = qkv[(int64_t)i * 3 * H + a * 64 + lane];
        wd[lane] = to_f32(dctx[(int64_t)i * H + a * 64 + lane]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        float sv[4], dp[4], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = lane + 64 * c;
            sv[c] = -INFINITY; dp[c] = 0.f;
            if (j < S) {
                float s = 0.f, t = 0.f;
                for (int d = 0; d < 64; ++d) { s = fmaf(wq[d], sA[j * 65 + d], s); t = fmaf(wd[d], sB[j * 65 + d], t); }
                s = s / 8.0f + (seen(i, j) ? 0.f : MODCR_NEG);
                sv[c] = s; dp[c] = p.side_post_drop ? (t + dalign(i, j)) * dmask(i, j) : t * dmask(i, j) + dalign(i, j);
                mx = fmaxf(mx, s);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (lane + 64 * c < S) { sv[c] = expf(sv[c] - mx); sum += sv[c]; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float dl = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (lane + 64 * c < S) { sv[c] *= inv; dl += sv[c] * dp[c]; }
        dl = wave_sum(dl);
#pragma unroll
        for (int c = 0; c < 4; ++c) if (lane + 64 * c < S) ws[lane + 64 * c] = sv[c] * (dp[c] - dl);
        if (lane == 0) { sMx[i] = mx; sInv[i] = inv; sDl[i] = dl; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        float g = 0.f;
        for (int j = 0; j < S; ++j) g = fmaf(ws[j], sA[j * 65 + lane], g);
        dqkv[(int64_t)i * 3 * H + a * 64 + lane] = g / 8.0f;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    __syncthreads();
    // ---- pass B ------------------------------------------------------------------------------------------
    for (int idx = tid; idx < S * 64; idx += 256) {
        const int i = idx >> 6, d = idx & 63;
        sA[i * 65 + d] = qkv[(int64_t)i * 3 * H + a * 64 + d];
        sB[i * 65 + d] = to_f32(dctx[(int64_t)i * H + a * 64 + d]);
    }
    __syncthreads();
    for (int j = wave; j < S; j += 4) {
        wq[lane] = qkv[(int64_t)j * 3 * H + H + a * 64 + lane];
        wd[lane] = qkv[(int64_t)j * 3 * H + 2 * H + a * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = lane + 64 * c;
            if (i < S) {
                float s = 0.f, t = 0.f;
                for (int d = 0; d < 64; ++d) { s = fmaf(sA[i * 65 + d], wq[d], s); t = fmaf(sB[i * 65 + d], wd[d], t); }
                s = s / 8.0f + (seen(i, j) ? 0.f : MODCR_NEG);
                const float pij = expf(s - sMx[i]) * sInv[i];
                const float mij = dmask(i, j);
                wp[i] = pij * mij;
                ws[i] = pij * ((p.side_post_drop ? (t + dalign(i, j)) * mij : t * mij + dalign(i, j)) - sDl[i]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        float gv = 0.f, gk = 0.f;
        for (int i = 0; i < S; ++i) { gv = fmaf(wp[i], sB[i * 65 + lane], gv); gk = fmaf(ws[i], sA[i * 65 + lane], gk); }
        dqkv[(int64_t)j * 3 * H + H + a * 64 + lane] = gk / 8.0f;
        dqkv[(int64_t)j * 3 * H + 2 * H + a * 64 + lane] = gv;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
}

}  // namespace

extern "C" int modcr_time_next_attn(void* start_event, void* stop_event) {
    g_time_start = (hipEvent_t)start_event;
    g_time_stop = (hipEvent_t)stop_event;
    return MODCR_OK;
}

// bytes of modcr_qkv_attn_lse_fwd's qkv_dump: [N][A][3][LP][64] bf16 with LP = the forward's token tile (0 = no dump at this S)
extern "C" int64_t modcr_qkv_attn_dump_bytes(int32_t N, int32_t S, int32_t A) {
    if (S <= 64 || S > 192) return 0;
    return (int64_t)N * A * 3 * (S <= 128 ? 128 : 192) * 64 * 2;
}

extern "C" int64_t modcr_qkv_attn_workspace(int32_t N, int32_t S, int32_t P, int32_t H, int32_t dtype) {
    if (dtype == MODCR_BF16) return P > 0 ? (int64_t)N * (P + S) * H * 2 : 0;      // [prefix ; x] rows for the tile kernels
    return (int64_t)N * (S + P) * 3 * H * (int64_t)sizeof(float);
}

extern "C" int modcr_qkv_attn_lse_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                                      const float* key_mask, const uint32_t* dense_mask_bits,
                                      const int32_t* chunk_id, int32_t chunk_t, void* ctx, float* probs,
                                      float* align_map, int32_t align_t, float* lse, void* qkv_dump, int32_t N, int32_t S, int32_t P,
                                      int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                                      void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);

extern "C" int modcr_qkv_attn_opt_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                                      const float* key_mask, const uint32_t* dense_mask_bits,
                                      const int32_t* chunk_id, int32_t chunk_t, void* ctx, float* probs,
                                      float* align_map, int32_t align_t, float* lse, void* qkv_dump, int32_t N, int32_t S, int32_t P,
                                      int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset, int32_t flags,
                                      void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);

extern "C" int modcr_qkv_attn_dropout_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                                          const float* key_mask, const uint32_t* dense_mask_bits,
                                          const int32_t* chunk_id, int32_t chunk_t, void* ctx, float* probs,
                                          float* align_map, int32_t align_t, int32_t N, int32_t S, int32_t P,
                                          int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                                          void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream) {
    return modcr_qkv_attn_lse_fwd(x, hist, wqkv, bqkv, key_mask, dense_mask_bits, chunk_id, chunk_t, ctx, probs, align_map, align_t, nullptr, nullptr,
                                  N, S, P, H, A, attn_p, seed, offset, workspace, workspace_bytes, dtype, stream);
}

extern "C" int modcr_qkv_attn_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                                  const float* key_mask, const uint32_t* dense_mask_bits,
                                  const int32_t* chunk_id, int32_t chunk_t, void* ctx, float* probs,
                                  float* align_map, int32_t align_t, int32_t N, int32_t S, int32_t P,
                                  int32_t H, int32_t A, void* workspace, int64_t workspace_bytes,
                                  int32_t dtype, modcr_stream_t stream) {
    return modcr_qkv_attn_dropout_fwd(x, hist, wqkv, bqkv, key_mask, dense_mask_bits, chunk_id, chunk_t, ctx, probs, align_map,
                                      align_t, N, S, P, H, A, 0.f, 0, 0, workspace, workspace_bytes, dtype, stream);
}

// lse (or NULL): [N, A, S] fp32, log2 of every query row's sum of exp2(log2e x score) -- the row statistics the five-product
// backward (modcr_qkv_attn_lse_bwd) rebuilds the probabilities from.  Written by the tile kernels only (bf16, 64 < S <= 256
// on their shapes): any other route returns MODCR_ERR_UNSUPPORTED when lse is given.
extern "C" int modcr_qkv_attn_lse_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                                      const float* key_mask, const uint32_t* dense_mask_bits,
                                      const int32_t* chunk_id, int32_t chunk_t, void* ctx, float* probs,
                                      float* align_map, int32_t align_t, float* lse, void* qkv_dump, int32_t N, int32_t S, int32_t P,
                                      int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                                      void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream) {
    return modcr_qkv_attn_opt_fwd(x, hist, wqkv, bqkv, key_mask, dense_mask_bits, chunk_id, chunk_t, ctx, probs, align_map, align_t, lse, qkv_dump,
                                  N, S, P, H, A, attn_p, seed, offset, 0, workspace, workspace_bytes, dtype, stream);
}

// flags: MODCR_ATTN_SIDE_POST_DROPOUT = the probabilities / align map leave AFTER the dropout, P o m / (1 - p), as the reference's
// modules return them (modeling_bert.py:69-74, v10:94-106); tile kernels only (bf16, 64 < P + S <= 256).  0 = the un-dropped ones.
extern "C" int modcr_qkv_attn_opt_fwd(const void* x, const void* hist, const void* wqkv, const float* bqkv,
                                      const float* key_mask, const uint32_t* dense_mask_bits,
                                      const int32_t* chunk_id, int32_t chunk_t, void* ctx, float* probs,
                                      float* align_map, int32_t align_t, float* lse, void* qkv_dump, int32_t N, int32_t S, int32_t P,
                                      int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset, int32_t flags,
                                      void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(x && wqkv && bqkv && ctx, "qkv_attn_fwd: null pointer");
    MODCR_REQUIRE((flags & ~MODCR_ATTN_SIDE_POST_DROPOUT) == 0, "qkv_attn_fwd: unknown flags 0x%x", flags);
    MODCR_REQUIRE(!qkv_dump || (lse && P == 0 && S > 64 && S <= 192 && modcr_aligned16(qkv_dump)),
                  "qkv_attn_fwd: the q|k|v dump comes with lse, without prefix rows, for 64 < S <= 192 (S=%d P=%d)", S, P);
    MODCR_REQUIRE(attn_p >= 0.f && attn_p < 1.f, "qkv_attn_fwd: attention dropout p=%g out of [0, 1)", attn_p);
    MODCR_REQUIRE(N > 0 && S > 0 && P >= 0 && A > 0, "qkv_attn_fwd: bad shape");
    MODCR_REQUIRE(H == A * 64, "qkv_attn_fwd: head size must be 64 (H=%d, A=%d)", H, A);
    MODCR_REQUIRE(P + S <= 256, "qkv_attn_fwd: P+S=%d exceeds 256 keys", P + S);
    MODCR_REQUIRE(P == 0 || hist, "qkv_attn_fwd: P=%d but hist is null", P);
    MODCR_REQUIRE(key_mask || dense_mask_bits, "qkv_attn_fwd: need key_mask or dense_mask_bits");
    MODCR_REQUIRE(!chunk_id || (chunk_t > 0 && chunk_t <= S), "qkv_attn_fwd: chunk_t=%d out of range", chunk_t);
    MODCR_REQUIRE(!align_map || (align_t > 0 && align_t < S), "qkv_attn_fwd: align_t=%d out of range", align_t);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MODCR_BF16) {
        MODCR_REQUIRE(modcr_aligned16(x) && modcr_aligned16(wqkv) && modcr_aligned16(ctx) &&
                          (P == 0 || modcr_aligned16(hist)),
                      "qkv_attn_fwd(bf16): 16-byte alignment");
        AttnArgs p;
        p.x = (const bf16*)x; p.hist = (const bf16*)hist; p.wqkv = (const bf16*)wqkv; p.bqkv = bqkv;
        p.key_mask = key_mask; p.bits = dense_mask_bits; p.chunk_id = chunk_id;
        p.ctx = (bf16*)ctx; p.probs = probs; p.align_map = align_map; p.lse = lse; p.dump = (bf16*)qkv_dump;
        p.N = N; p.S = S; p.P = P; p.H = H; p.A = A; p.chunk_t = chunk_t; p.align_t = align_t;
        p.drop_thr2 = 0; p.drop_on = 0; p.drop_key = 0; p.drop_keep = 1.f;
        p.side_post_drop = (attn_p > 0.f && (flags & MODCR_ATTN_SIDE_POST_DROPOUT)) ? 1 : 0;
        if (attn_p > 0.f) {
            // the reference returns the probabilities AFTER the dropout (modeling_bert.py:74): a probabilities output comes with
            // MODCR_ATTN_SIDE_POST_DROPOUT (the generic tile variant applies the mask to it); the un-dropped ones are not offered
            if (probs && !p.side_post_drop) {
                modcr_set_error("qkv_attn_fwd: a probabilities output under attention-probability dropout needs MODCR_ATTN_SIDE_POST_DROPOUT");
                return MODCR_ERR_UNSUPPORTED;
            }
            p.drop_key = seed + offset * 0x9E3779B97F4A7C15ull;
            p.drop_thr2 = attn_thrm1_2(attn_thr16(attn_p)); p.drop_on = 1;
            p.drop_keep = 1.0f / (1.0f - attn_p);
        }
        // knobs below: tuning build only (common.h); the product library takes the defaults
        p.debug = modcr_knob_int("MODCR_ATTN_DEBUG", 0);
#ifdef MODCR_TUNING
        if (getenv("MODCR_ATTN_TRACE_PTR")) p.trace = reinterpret_cast<unsigned long long*>(strtoull(getenv("MODCR_ATTN_TRACE_PTR"), nullptr, 0));
#endif
        p.hconc = modcr_knob_int("MODCR_ATTN_HCONC", 3);     // tile kernels: 3 head pairs at a time per XCD (measured 166 vs 168 us; FETCH_SIZE: profiles/)
        const int L = P + S;
        const int one_head = modcr_knob_set("MODCR_ATTN_HPW1");
        [[maybe_unused]] const int ring64 = modcr_knob_set("MODCR_ATTN_RING64");          // 64-wide K-tiles, 2 slots
        const bool pair = (A % 2 == 0) && !one_head;
        // prefix rows (history_state) on the tile kernels: K and V over [prefix ; x] (modeling_bert.py:36-44), queries from x
        // only.  The caller's workspace receives the concatenated rows (one copy pass); key-mask / dense-mask calls without
        // side outputs take this route, everything else the older kernel.
        bool prefix_tiles = false;
        if (P > 0 && L > 64 && L <= 256 && (pair || L > 192) && !probs && !align_map && !chunk_id && (H % 128) == 0 && H >= 256 &&
            workspace && workspace_bytes >= modcr_qkv_attn_workspace(N, S, P, H, dtype) && modcr_aligned16(workspace) &&
            (int64_t)3 * H * H * 2 < (1ll << 31) && !modcr_knob_set("MODCR_ATTN_NO_PREFIX_TILES") &&
            !modcr_knob_set("MODCR_ATTN_NO_V4") && !modcr_knob_set("MODCR_ATTN_NO_V4S")) {
            const int64_t pieces = (int64_t)N * L * (H >> 3);
            const int grid = (int)((pieces + 255) / 256 < 16384 ? (pieces + 255) / 256 : 16384);
            hipLaunchKernelGGL(concat_prefix_kernel, dim3(grid), dim3(256), 0, st, p.hist, p.x, (bf16*)workspace, N, P, S, H);
            int rc = modcr_check_launch("concat_prefix");
            if (rc != MODCR_OK) return rc;
            p.x = (const bf16*)workspace;
            prefix_tiles = true;
        }
        // (row statistics come from the tile kernels only)
#define MODCR_NO_LSE_HERE() do { if (lse) { modcr_set_error("qkv_attn_fwd: no row statistics (lse) on this route (S=%d P=%d A=%d H=%d)", S, P, A, H); return MODCR_ERR_UNSUPPORTED; } \
                                 if (p.side_post_drop && (probs || align_map)) { modcr_set_error("qkv_attn_fwd: post-dropout side outputs on the tile kernels only (S=%d P=%d A=%d H=%d)", S, P, A, H); return MODCR_ERR_UNSUPPORTED; } } while (0)
        if (L <= 64) MODCR_NO_LSE_HERE();
        if (L <= 64) return pair ? launch_attn<2, 2, 2, 64, 2>(p, st) : launch_attn<2, 1, 2, 64, 2>(p, st);
        if (L <= 128 && L > 64 && (A % 2 == 0) && !one_head && (P == 0 || prefix_tiles) && (H % 128) == 0 && H >= 256 && (int64_t)3 * H * H * 2 < (1ll << 31)) {
            // 64 < S <= 128: the same kernel on a 128-token tile (A half = 64 rows = one LDS-DMA piece per wave)
            const int no_v4s = modcr_knob_set("MODCR_ATTN_NO_V4S");
            if (!no_v4s) {
                if (!probs && align_map && chunk_id && dense_mask_bits) return launch_attn4<3, 128>(p, st);
                if (probs || align_map || chunk_id) return launch_attn4<0, 128>(p, st);
                return dense_mask_bits ? launch_attn4<2, 128>(p, st) : launch_attn4<1, 128>(p, st);
            }
        }
        if (L <= 128) {
            MODCR_NO_LSE_HERE();
#ifdef MODCR_TUNING             // (A/B variants of the older kernel: instantiated in the tuning library only)
            const int ring32 = modcr_knob_int("MODCR_ATTN_RING32", 0);
            if (pair && ring32 == 1) return launch_attn<4, 2, 2, 32, 4>(p, st);
            if (pair && ring32 == 2) return launch_attn<4, 2, 2, 32, 3>(p, st);
#endif
            return pair ? launch_attn<4, 2, 2, 64, 2>(p, st) : launch_attn<4, 1, 2, 64, 2>(p, st);
        }
        if (L <= 192) {
            const int no_v4 = modcr_knob_set("MODCR_ATTN_NO_V4");
            if (pair && !no_v4 && L > 128 && (P == 0 || prefix_tiles) && (H % 128) == 0 && H >= 256 && (int64_t)3 * H * H * 2 < (1ll << 31)) {
                if (!probs && align_map && chunk_id && dense_mask_bits) return launch_attn4<3, 192>(p, st);
                if (probs || align_map || chunk_id) return launch_attn4<0, 192>(p, st);
                return dense_mask_bits ? launch_attn4<2, 192>(p, st) : launch_attn4<1, 192>(p, st);
            }
            MODCR_NO_LSE_HERE();
            if (!pair) {
#ifdef MODCR_TUNING
                const int v = modcr_knob_int("MODCR_ATTN_HPW1", 0);
                if (v == 2) return launch_attn<6, 1, 3, 32, 3>(p, st);   // 2 workgroups per CU
                if (v == 3) return launch_attn<6, 1, 3, 32, 2>(p, st);
#endif
                return launch_attn<6, 1, 2, 64, 2>(p, st);
            }
#ifdef MODCR_TUNING
            if (ring64) return launch_attn<6, 2, 3, 64, 2>(p, st);
#endif
            return launch_attn<6, 2, 3, 32, 4>(p, st);
        }
        // 192 < L <= 256 (the VCR / Oscar-large shape class S = 230): the 256-token tile, one head per workgroup; key-mask and
        // dense-mask calls without side outputs, and the phase-3 call (dense mask + chunk-mean queries + align map) when its
        // [T][R] fp32 tile fits the one V^T image (VCR: 194 x 36); everything else: the older kernel
        if ((P == 0 || prefix_tiles) && !probs && (H % 128) == 0 && H >= 256 && (int64_t)3 * H * H * 2 < (1ll << 31) &&
            !modcr_knob_set("MODCR_ATTN_NO_V4L")) {
            if (!align_map && !chunk_id) return dense_mask_bits ? launch_attn4<2, 256, 1>(p, st) : launch_attn4<1, 256, 1>(p, st);
            if (align_map && chunk_id && dense_mask_bits && P == 0 &&
                (int64_t)align_t * (S - align_t) * 4 <= (int64_t)64 * A4T<256, 1>::VT_STRIDE)
                return launch_attn4<3, 256, 1>(p, st);
        }
        // a probabilities output, or chunk-mean queries / an align map in any other combination (incl. a [T][R] tile larger than the
        // V^T image): the generic variant of the same tile (exact pass; its align tile lies over K | V^T)
        if (P == 0 && (H % 128) == 0 && H >= 256 && (int64_t)3 * H * H * 2 < (1ll << 31) && !modcr_knob_set("MODCR_ATTN_NO_V4L"))
            return launch_attn4<0, 256, 1>(p, st);
        MODCR_NO_LSE_HERE();
        return launch_attn<8, 1, 2, 64, 2>(p, st);
    }
    MODCR_REQUIRE(!lse, "qkv_attn_fwd(f32): row statistics (lse) are written on the bf16 path only");
    MODCR_REQUIRE(dtype == MODCR_F32, "qkv_attn_fwd: unknown dtype %d", dtype);
    MODCR_REQUIRE(attn_p == 0.f || !probs || (flags & MODCR_ATTN_SIDE_POST_DROPOUT),
                  "qkv_attn_fwd(f32): a probabilities output under attention-probability dropout needs MODCR_ATTN_SIDE_POST_DROPOUT");
    const int64_t need = modcr_qkv_attn_workspace(N, S, P, H, dtype);
    MODCR_REQUIRE(workspace && workspace_bytes >= need, "qkv_attn_fwd(f32): workspace %lld < %lld bytes",
                  (long long)workspace_bytes, (long long)need);
    float* qkv_x = (float*)workspace;
    float* qkv_h = qkv_x + (int64_t)N * S * 3 * H;
    int rc = modcr_linear_fwd(x, H, wqkv, H, bqkv, nullptr, 0, 0, qkv_x, 3 * H, N * S, 3 * H, H,
                              MODCR_ACT_NONE, MODCR_F32, MODCR_F32, stream);
    if (rc != MODCR_OK) return rc;
    if (P > 0) {
        rc = modcr_linear_fwd(hist, H, wqkv, H, bqkv, nullptr, 0, 0, qkv_h, 3 * H, N * P, 3 * H, H,
                              MODCR_ACT_NONE, MODCR_F32, MODCR_F32, stream);
        if (rc != MODCR_OK) return rc;
    }
    if (chunk_id) {
        rc = modcr_chunk_mean_q_fwd(qkv_x, 3 * H, (int64_t)S * 3 * H, chunk_id, N, chunk_t, H, MODCR_F32, stream);
        if (rc != MODCR_OK) return rc;
    }
    AttnF32Args f;
    f.qkv_x = qkv_x; f.qkv_h = P > 0 ? qkv_h : nullptr; f.key_mask = key_mask; f.bits = dense_mask_bits;
    f.ctx = (float*)ctx; f.probs = probs; f.align_map = align_map;
    f.N = N; f.S = S; f.P = P; f.H = H; f.A = A; f.align_t = align_t;
    f.drop_thr16 = 0; f.drop_key = 0; f.drop_keep = 1.f; f.side_post_drop = 0;
    if (attn_p > 0.f) {                                     // exact-parity route: the mask of the bf16 kernels, so the two routes can be compared
        f.drop_thr16 = attn_thr16(attn_p); f.drop_key = seed + offset * 0x9E3779B97F4A7C15ull;
        f.drop_keep = 1.0f / (1.0f - attn_p); f.side_post_drop = (flags & MODCR_ATTN_SIDE_POST_DROPOUT) ? 1 : 0;
    }
    const int L = P + S;
    const size_t smem = ((size_t)2 * L * 65 + 256 + 4 * (size_t)L) * sizeof(float);
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            modcr_set_error("qkv_attn(f32): cannot reserve LDS: %s", hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    hipLaunchKernelGGL(attn_f32_kernel, dim3(N * A), dim3(256), smem, st, f);
    return modcr_check_launch("attn_f32");
}


// ---- backward of modcr_qkv_attn_fwd (no prefix rows) ---------------------------------------------------------------
static int64_t attn_bwd_sub_ws(int32_t M, int32_t H) {
    const int64_t a = modcr_linear_bwd_input_workspace(M, 3 * H, H), b = modcr_linear_bwd_weight_workspace(M, 3 * H, H);
    return ((a > b ? a : b) + 255) & ~(int64_t)255;
}
extern "C" int64_t modcr_qkv_attn_bwd_workspace(int32_t N, int32_t S, int32_t H, int32_t dtype) {
    (void)dtype;
    const int64_t rows = (int64_t)N * S * 3 * H * (int64_t)sizeof(float);
    return 2 * ((rows + 255) & ~(int64_t)255) + attn_bwd_sub_ws(N * S, H);
}

extern "C" int modcr_qkv_attn_lse_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                                      const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                                      int32_t chunk_t, const float* dx_residual, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                                      int32_t S, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                                      const float* d_align, int32_t align_t, const void* ctx, const float* lse, const void* qkv_dump,
                                      void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);

extern "C" int modcr_qkv_attn_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                                  const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                                  int32_t chunk_t, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                                  int32_t S, int32_t H, int32_t A, void* workspace, int64_t workspace_bytes,
                                  int32_t dtype, modcr_stream_t stream) {
    return modcr_qkv_attn_lse_bwd(dctx, x, wqkv, bqkv, key_mask, dense_mask_bits, chunk_id, chunk_t, nullptr, dx, dwqkv, dbqkv, accumulate,
                                  N, S, H, A, 0.f, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, workspace, workspace_bytes, dtype, stream);
}

extern "C" int modcr_qkv_attn_dropout_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                                          const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                                          int32_t chunk_t, const float* dx_residual, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                                          int32_t S, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                                          const float* d_align, int32_t align_t,
                                          void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream) {
    return modcr_qkv_attn_lse_bwd(dctx, x, wqkv, bqkv, key_mask, dense_mask_bits, chunk_id, chunk_t, dx_residual, dx, dwqkv, dbqkv, accumulate,
                                  N, S, H, A, attn_p, seed, offset, d_align, align_t, nullptr, nullptr, nullptr, workspace, workspace_bytes, dtype, stream);
}

// ctx + lse (both or neither; bf16 path): the forward's context rows and the row statistics modcr_qkv_attn_lse_fwd wrote.  With
// them the attention core runs as the five-product kernel of attn_bwd.hip (P rebuilt from lse, delta = rowsum(dO o O));
// without them, or with an align-map gradient, the older core recomputes the statistics (eight products).
extern "C" int modcr_qkv_attn_opt_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                                      const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                                      int32_t chunk_t, const float* dx_residual, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                                      int32_t S, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                                      const float* d_align, int32_t align_t, const void* ctx, const float* lse, const void* qkv_dump, int32_t flags,
                                      void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream);

extern "C" int modcr_qkv_attn_lse_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                                      const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                                      int32_t chunk_t, const float* dx_residual, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                                      int32_t S, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                                      const float* d_align, int32_t align_t, const void* ctx, const float* lse, const void* qkv_dump,
                                      void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream) {
    return modcr_qkv_attn_opt_bwd(dctx, x, wqkv, bqkv, key_mask, dense_mask_bits, chunk_id, chunk_t, dx_residual, dx, dwqkv, dbqkv, accumulate,
                                  N, S, H, A, attn_p, seed, offset, d_align, align_t, ctx, lse, qkv_dump, 0, workspace, workspace_bytes, dtype, stream);
}

// flags: MODCR_ATTN_SIDE_POST_DROPOUT = d_align is the gradient of an align map that summed the probabilities AFTER the dropout
// (modcr_qkv_attn_opt_fwd with the same flag): it enters dP under the forward's mask, m / (1 - p) o d_align
extern "C" int modcr_qkv_attn_opt_bwd(const void* dctx, const void* x, const void* wqkv, const float* bqkv,
                                      const float* key_mask, const uint32_t* dense_mask_bits, const int32_t* chunk_id,
                                      int32_t chunk_t, const float* dx_residual, void* dx, float* dwqkv, float* dbqkv, int32_t accumulate, int32_t N,
                                      int32_t S, int32_t H, int32_t A, float attn_p, uint64_t seed, uint64_t offset,
                                      const float* d_align, int32_t align_t, const void* ctx, const float* lse, const void* qkv_dump, int32_t flags,
                                      void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(dctx && x && wqkv && bqkv && dx && dwqkv && dbqkv, "qkv_attn_bwd: null pointer");
    MODCR_REQUIRE((flags & ~MODCR_ATTN_SIDE_POST_DROPOUT) == 0, "qkv_attn_bwd: unknown flags 0x%x", flags);
    MODCR_REQUIRE(!d_align || (align_t > 0 && align_t < S), "qkv_attn_bwd: align_t=%d out of range", align_t);
    MODCR_REQUIRE((ctx == nullptr) == (lse == nullptr), "qkv_attn_bwd: ctx and lse come together");
    MODCR_REQUIRE(!qkv_dump || (lse && dtype == MODCR_BF16 && S > 64 && S <= 192),
                  "qkv_attn_bwd: the forward's q|k|v dump is used with ctx + lse, bf16, 64 < S <= 192");
    MODCR_REQUIRE(attn_p >= 0.f && attn_p < 1.f, "qkv_attn_bwd: attention dropout p=%g out of [0, 1)", attn_p);
    // the mask of modcr_qkv_attn_dropout_fwd exists for its tile kernels only, and only the MFMA core regenerates it
    // (64 < S <= 192: the MFMA cores; 192 < S <= 256, the forward's 256-token tile: the exact core below -- the shape class of
    // BASELINE config 5 with trainable encoders, not a benched path)
    // (every core regenerates the forward's mask from (seed, offset): the MFMA cores at 64 < S <= 192, the exact core elsewhere and in fp32)
    MODCR_REQUIRE(attn_p == 0.f || S <= 256, "qkv_attn_bwd: attention-probability dropout needs S <= 256 (S=%d)", S);
    MODCR_REQUIRE(N > 0 && S > 0 && S <= 256 && A > 0 && H == A * 64, "qkv_attn_bwd: bad shape (N=%d S=%d H=%d A=%d)", N, S, H, A);
    MODCR_REQUIRE(key_mask || dense_mask_bits, "qkv_attn_bwd: need key_mask or dense_mask_bits");
    MODCR_REQUIRE(dtype == MODCR_BF16 || dtype == MODCR_F32, "qkv_attn_bwd: unknown dtype %d", dtype);
    const int64_t need = modcr_qkv_attn_bwd_workspace(N, S, H, dtype);
    MODCR_REQUIRE(workspace && workspace_bytes >= need, "qkv_attn_bwd: workspace %lld < %lld bytes",
                  (long long)workspace_bytes, (long long)need);
    const int M = N * S;
    const int64_t rows = (((int64_t)M * 3 * H * (int64_t)sizeof(float)) + 255) & ~(int64_t)255;
    float* qkv = (float*)workspace;
    float* dqkv = (float*)((char*)workspace + rows);
    // bf16: the two weight / input products run on the MFMA path (workspace); fp32 parity path: exact VALU kernels
    void* sub = dtype == MODCR_BF16 ? (void*)((char*)workspace + 2 * rows) : nullptr;
    const int64_t sub_bytes = dtype == MODCR_BF16 ? attn_bwd_sub_ws(M, H) : 0;
    // 1. recompute q | k | v rows, chunk-mean queries as the forward.  MFMA core: bf16 rows (what the forward's LDS images held;
    //    half the bytes of the fp32 rows round 1 wrote and read back: 849 MB per call at 128 examples); exact core: fp32
    const bool mfma_core = dtype == MODCR_BF16 && S <= AB::LP && !modcr_knob_set("MODCR_ATTN_BWD_VALU");
    const int32_t qdt = mfma_core ? MODCR_BF16 : MODCR_F32;
    int rc = MODCR_OK;
    if (!qkv_dump) {                // (with the forward's dump nothing is recomputed: its Q image already holds the chunk means)
        rc = modcr_linear_fwd(x, H, wqkv, H, bqkv, nullptr, 0, 0, qkv, 3 * H, M, 3 * H, H, MODCR_ACT_NONE, dtype, qdt, stream);
        if (rc != MODCR_OK) return rc;
        if (chunk_id) {
            rc = modcr_chunk_mean_q_fwd(qkv, 3 * H, (int64_t)S * 3 * H, chunk_id, N, chunk_t, H, qdt, stream);
            if (rc != MODCR_OK) return rc;
        }
    }
    // 2. attention core backward
    AttnBwdArgs b;
    b.qkv = qkv; b.qkvb = reinterpret_cast<const bf16*>(qkv); b.dctx = dctx; b.key_mask = key_mask; b.bits = dense_mask_bits; b.dqkv = dqkv;
    b.N = N; b.S = S; b.H = H; b.A = A; b.out_bf16 = 0;
    b.drop_thr16 = 0; b.drop_key = 0; b.drop_keep = 1.f;
    b.side_post_drop = (attn_p > 0.f && (flags & MODCR_ATTN_SIDE_POST_DROPOUT)) ? 1 : 0;
    b.d_align = d_align; b.align_t = align_t;
    b.ctx = reinterpret_cast<const bf16*>(ctx); b.lse = lse; b.dump = reinterpret_cast<const bf16*>(qkv_dump);
    b.delta_align = qkv_dump ? qkv : nullptr;               // (with the dump the q|k|v area of the workspace is free: N A S floats of it)
    b.debug = modcr_knob_int("MODCR_ATTN_BWD_DEBUG", 0);                 // tuning build only
    if (attn_p > 0.f) {
        b.drop_key = seed + offset * 0x9E3779B97F4A7C15ull;
        b.drop_thr16 = attn_thr16(attn_p);                  // (as the forward)
        b.drop_keep = 1.0f / (1.0f - attn_p);
    }
    const size_t smem = ((size_t)2 * S * 65 + 3 * (size_t)S + 4 * (128 + 2 * (size_t)S)) * sizeof(float);
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_f32_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_f32_kernel<bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        configured = true;
    }
    int gdt = MODCR_F32;            // dtype of the dq | dk | dv rows
    if (mfma_core && lse && (!d_align || qkv_dump) && !modcr_knob_set("MODCR_ATTN_BWD_OLD")) {
        b.out_bf16 = 1; gdt = MODCR_BF16;
        rc = modcr_launch_attn_bwd5(b, (hipStream_t)stream);
        if (rc != MODCR_OK) return rc;
    } else if (mfma_core) {
        b.out_bf16 = 1; gdt = MODCR_BF16;
        static bool configured2_dev[MODCR_MAX_DEV] = {};
        bool& configured2 = configured2_dev[modcr_device_index()];
        if (!configured2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_mfma_kernel<bf16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, AB::SMEM);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_mfma_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, AB::SMEM);
            configured2 = true;
        }
        if (d_align) hipLaunchKernelGGL((attn_bwd_mfma_kernel<bf16, true>), dim3(N * A), dim3(AB::NT), AB::SMEM, (hipStream_t)stream, b);
        else hipLaunchKernelGGL((attn_bwd_mfma_kernel<bf16, false>), dim3(N * A), dim3(AB::NT), AB::SMEM, (hipStream_t)stream, b);
    } else if (dtype == MODCR_BF16) hipLaunchKernelGGL(attn_bwd_f32_kernel<bf16>, dim3(N * A), dim3(256), smem, (hipStream_t)stream, b);
    else hipLaunchKernelGGL(attn_bwd_f32_kernel<float>, dim3(N * A), dim3(256), smem, (hipStream_t)stream, b);
    rc = modcr_check_launch("attn_bwd");
    if (rc != MODCR_OK) return rc;
    // 3. the chunk mean is its own adjoint: dq rows of a chunk <- their mean (v10:66-78)
    if (chunk_id) {
        rc = modcr_chunk_mean_q_fwd(dqkv, 3 * H, (int64_t)S * 3 * H, chunk_id, N, chunk_t, H, gdt, stream);
        if (rc != MODCR_OK) return rc;
    }
    // 4. dX = dqkv . Wqkv;  dWqkv (+)= dqkv^T . X;  dbqkv (+)= column sums
    rc = modcr_linear_bwd_input_res(dqkv, 3 * H, gdt, wqkv, H, dx_residual, H, dx, H, M, 3 * H, H, dtype, dtype, sub, sub_bytes, stream);
    if (rc != MODCR_OK) return rc;
    return modcr_linear_bwd_weight(dqkv, 3 * H, gdt, x, H, dwqkv, dbqkv, M, 3 * H, H, accumulate, dtype, sub, sub_bytes, stream);
}
