// nn.Linear on CDNA4:  C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual)
//
//  * bf16 path: 128x128x64 workgroup tile, 4 waves (2x2), each wave 64x64 = 2x2 tiles of
//    v_mfma_f32_32x32x16_bf16.  Both operands are K-contiguous in HBM (activations row-major,
//    weights [out][in]), so a lane's MFMA fragment is one 16-byte chunk of a row: the tiles are
//    staged through LDS as 128-byte rows whose 16-byte chunks are XOR-swizzled by (row>>1)&7 so a
//    ds_read_b128 lane group (16 distinct rows, same chunk) covers all 16 slots of the 256-byte
//    bank row.  Tiles arrive by LDS-DMA (global_load_lds_dwordx4, swizzle applied to the source
//    address) into a double buffer: the next tile is in flight while the current one feeds the
//    MFMAs; one barrier per K-tile.
//  * fp32 path: plain 64x64x16 VALU tile with arbitrary element strides (parity path and the
//    transposed products of the head backward); exact fp32 fma chains.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BK = 64;

// 2-byte output types: MODCR_BF16, or MODCR_F16 (IEEE half: the sublayer output / pre-LayerNorm rows that a LayerNorm pass
// reads back -- three more mantissa bits than bf16 at the same bytes, so the round trip costs no accuracy at bf16 output
// precision where a bf16 round trip measurably did)
template <int OUT> struct Out16 { typedef bf16 T; };
template <> struct Out16<MODCR_F16> { typedef _Float16 T; };
template <int OUT> using o16x4 = typename Out16<OUT>::T __attribute__((ext_vector_type(4)));
template <int OUT> using o16x8 = typename Out16<OUT>::T __attribute__((ext_vector_type(8)));
// accumulator -> 16-bit output element.  IEEE half saturates at +-65504 instead of overflowing to inf (a pre-LayerNorm row with one
// outlier feature would otherwise come out of the LayerNorm pass as NaN; bf16 has fp32's range and needs nothing): one v_med3_f32.
template <int OUT> __device__ __forceinline__ typename Out16<OUT>::T cvt16(float v) {
    // (v_med3_f32 returns the MINIMUM when an input is NaN: a NaN accumulator must stay NaN, not become -65504, or a diverged run
    // produces finite garbage; fminf / fmaxf lower to v_min_f32 / v_max_f32 which would drop the NaN too, hence the select)
    if constexpr (OUT == MODCR_F16) v = (v == v) ? __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f) : v;
    return (typename Out16<OUT>::T)v;
}

struct LinearArgs {
    const bf16* A; int64_t lda;
    const bf16* W; int64_t ldw;
    const float* bias;
    const void* res; int64_t ldr; int res_dtype;
    void* C; int64_t ldc; int out_dtype;
    void* C2 = nullptr;             // MODCR_ACT_GELU_KEEP: bf16 [M,N] (row stride ldc), the pre-activation values
    int M, N, K, act;
    int tiles_m, tiles_n, vec_ok;
    int ngroup;     // persistent 256 x 256 kernel: column tiles per group of the tile walk (0 = row-major walk), see launch_p8d
    int kvalid = 0; // TN = 2 (token-major W operand): tokens that exist in W; K-tiles past them re-read the last valid one (their A columns are zero)
    int k_tiles_per_split;          // split-K: blockIdx.y owns K-tiles [y*kps, (y+1)*kps); 0 = no split
    int64_t split_stride;           // elements between the partial outputs of consecutive splits
    int order;                      // tuning build only (MODCR_GEMM_ORDER, compiled out of the product library): bit0 = column-major tile order, bit1 = no XCD remap
    int rev_walk = 0;               // persistent kernels: walk the tiles from the last to the first
    int stpol = 0;                  // tuning build only (MODCR_GEMM_STPOL): cache-policy bits of the seamless-ring epilogue's stores
    int pf_next = 0;                // tuning build only (MODCR_GEMM_PF): the seamless-ring kernel's epilogue touches the NEXT tile's activation rows (see epilogue_spec)
    int trace_wg = 0;               // tuning build only (MODCR_GEMM_TRACE_WG): the workgroup whose tile seams are stamped
};

__device__ __forceinline__ int swz(int row, int chunk) { return (row << 7) + (((chunk ^ (row >> 1)) & 7) << 4); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Tile shapes.  The per-CU L2->LDS fill rate (~70 GB/s/CU measured for LDS-DMA) bounds a tile's
// MFMA rate through its arithmetic intensity: 128x128 = 64 FLOP per staged byte, 256x256 = 128.
//   Tile<128,128,2,2>: 4 waves, wave tile 64x64,  64 KB LDS -> two workgroups per CU (small / ragged shapes)
//   Tile<256,256,2,4>: 8 waves, wave tile 128x64, 128 KB LDS -> one workgroup per CU (the big GEMMs)
template <int BM_, int BN_, int WM_, int WN_, int BKT_, int NSLOT_, int ASLOT_ = NSLOT_>
struct Tile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BKT = BKT_, NSLOT = NSLOT_, ASLOT = ASLOT_;
    static constexpr int NWAVES = WM * WN, NT = NWAVES * 64;
    static constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);      // 32x32 MFMA tiles per wave
    static constexpr int ROWB = BKT * 2;                                // bytes per staged row (128 or 64)
    static constexpr int RPC = 1024 / ROWB;                             // rows per 1-KiB DMA chunk
    static constexpr int CPR = ROWB / 16;                               // 16-byte chunks per row
    static constexpr int STAGE = (BM + BN) * ROWB;                      // A tile then W tile
    static constexpr bool PINGPONG = (WM == 2 && WM * WN == 8 && BKT == 64 && NSLOT == 2);   // see the K loop
    static constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB;
    // ring of K-tiles; the ping-pong tiles keep separate rings for A (ASLOT deep) and W (2 deep)
    static constexpr int SMEM = PINGPONG ? ASLOT * A_BYTES + 2 * W_BYTES : NSLOT * STAGE;
    static constexpr int CPW = (BM + BN) / RPC / NWAVES;                // DMA chunks per wave per K-tile
    static constexpr int slab_rows() {                                   // epilogue rows per pass: fp32 [SLAB][BN] in the ring
        int best = 32;
        for (int c = 32; c <= BM; c += 32)
            if (BM % c == 0 && c * BN * 4 <= SMEM) best = c;
        return best;
    }
    static constexpr int SLAB = slab_rows();
    static_assert(BKT == 64 || BKT == 32, "K-tile width");
    static_assert((BM + BN) / RPC % NWAVES == 0, "DMA chunks must divide over the waves");
    static_assert(BM % SLAB == 0 && SLAB % 32 == 0 && SLAB * BN * 4 <= SMEM, "epilogue slab");
    static_assert((NSLOT - 1) * CPW < 64, "vmcnt field");
    // byte offset of logical 16-byte chunk `chunk` of row `row`: XOR swizzle so that a ds_read_b128
    // lane group (16 distinct rows, same logical chunk) covers all 16 slots of the 256-byte bank row
    static __device__ __forceinline__ int off(int row, int chunk) {
        return BKT == 64 ? (row << 7) + (((chunk ^ (row >> 1)) & 7) << 4) : (row << 6) + (((chunk ^ (row >> 2)) & 3) << 4);
    }
    static __device__ __forceinline__ int key(int row) { return BKT == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }
};

// RES: 0 = no residual, 1 = bf16 residual, 2 = fp32 residual.  OUT: MODCR_BF16 / MODCR_F32.
template <typename T, int ACT, int RES, int OUT>
__global__ __launch_bounds__(T::NT) void linear_bf16_kernel(LinearArgs p) {
    constexpr int BM = T::BM, BN = T::BN, TM = T::TM, TN = T::TN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = MODCR_DBG(p.order & 2) ? (int)blockIdx.x : xcd_remap(blockIdx.x, nwg);
    const int tm = MODCR_DBG(p.order & 1) ? tile % p.tiles_m : tile / p.tiles_n;
    const int tn = MODCR_DBG(p.order & 1) ? tile / p.tiles_m : tile % p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WN, wn = wave % T::WN;
    const int r = lane & 31, h = lane >> 5;

    // LDS-DMA staging: chunk c (1 KiB = RPC rows, one instruction) of the [A tile ; W tile] image
    // goes to wave c % NWAVES.  LDS is written linearly (base + 16*lane); the XOR swizzle is
    // applied to the SOURCE chunk so that slot s of row r holds logical chunk s ^ key(r).
    const bf16* gsrc[T::CPW];
#pragma unroll
    for (int q = 0; q < T::CPW; ++q) {
        const int row = (wave + q * T::NWAVES) * T::RPC + lane / T::CPR;   // row of the stacked image
        const int c = (lane % T::CPR) ^ T::key(row);
        gsrc[q] = (row < BM) ? p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + c * 8
                             : p.W + (int64_t)min(n0 + row - BM, p.N - 1) * p.ldw + c * 8;
    }
    auto stage = [&](int slot, int k0) {
#pragma unroll
        for (int q = 0; q < T::CPW; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[q] + k0),
                                             (lptr_t)(smem + slot * T::STAGE + (wave + q * T::NWAVES) * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // K loop over a ring of NSLOT tiles with NSLOT-1 tiles in flight: the per-CU L2->LDS fill is
    // latency-bound (~70 GB/s with one tile in flight), so bytes in flight are what buys bandwidth.
    // Top of iteration kt: wait (counted vmcnt) until tile kt has landed, raw barrier, refill the
    // slot everyone finished reading one iteration ago, compute.  No vmcnt(0) until the tail.
    constexpr int PRE = T::NSLOT - 1;
    constexpr int KSTEPS = T::BKT / 16;
    const int nk_all = p.K / T::BKT;
    const int kps = p.k_tiles_per_split * (BK / T::BKT);            // split sizes are given in 64-wide tiles
    const int kt0 = kps ? blockIdx.y * kps : 0;
    const int nk = MODCR_DBG(p.order & 4) ? 0 : (kps ? max(0, min(nk_all - kt0, kps)) : nk_all);   // bit2: timing-only, skip the K loop
    if (kps) p.C = reinterpret_cast<float*>(p.C) + (int64_t)blockIdx.y * p.split_stride;
    if constexpr (T::PINGPONG) {
        // 8 waves = two groups of four (group = wm), one wave of each group per SIMD.  The groups run
        // the same program ONE BARRIER APART: while group 0 issues a block of MFMAs, group 1 does its
        // LDS fragment reads / LDS-DMA refill for its next block, then they swap.  Every SIMD's matrix
        // pipe is fed by one of its two waves at all times; the barriers enforce the alternation.
        //   per K-tile and group:  LOAD(k-steps 0,1) | MFMA | LOAD(k-steps 2,3) | MFMA      (4 barriers)
        // Ring hazards (2 slots): tile kt+1 is DMA'd into the slot of tile kt-1 during LOAD(kt,0), i.e.
        // after the barrier that follows the lagging group's last read of tile kt-1; every wave drains
        // its own DMA (vmcnt(0)) in the phase that ends at the barrier in front of the leading group's
        // first read of tile kt+1.
        const int g = wm;
        bf16x8 fa[2][TM], fb[2][TN];
        auto load_frags = [&](const unsigned char* sA, const unsigned char* sB, int set, int ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[set][i] = *reinterpret_cast<const bf16x8*>(sA + T::off((wm * TM + i) * 32 + r, ks * 2 + h));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[set][j] = *reinterpret_cast<const bf16x8*>(sB + T::off((wn * TN + j) * 32 + r, ks * 2 + h));
        };
        auto mfma_block = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int set = 0; set < 2; ++set)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        };
        // A and W have their own rings.  W (re-read by every M-tile, L2/MALL resident) is refilled one
        // K-tile ahead; A is streamed once from HBM and is kept ASLOT-1 K-tiles ahead: measured, a
        // wave sustains ~7 GB/s of LDS-DMA and first-touch rows take ~2 us, so one tile of A in
        // flight per CU cannot keep the matrix pipe fed.  Issue order W then A, so the counted
        // vmcnt(QA) at the end of a tile waits for everything except the youngest A tile.
        constexpr int QA = BM / 8 / T::NWAVES, QW = BN / 8 / T::NWAVES;
        static_assert(BM / 8 % T::NWAVES == 0 && QA + QW == T::CPW, "A / W pieces per wave");
        unsigned char* const ringA = smem;
        unsigned char* const ringW = smem + T::ASLOT * T::A_BYTES;
        auto stage_A = [&](int slot, int k0) {
#pragma unroll
            for (int q = 0; q < QA; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[q] + k0),
                                                 (lptr_t)(ringA + slot * T::A_BYTES + (wave + q * T::NWAVES) * 1024), 16, 0, 0);
        };
        auto stage_W = [&](int slot, int k0) {
#pragma unroll
            for (int q = 0; q < QW; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[QA + q] + k0),
                                                 (lptr_t)(ringW + slot * T::W_BYTES + (wave + q * T::NWAVES) * 1024), 16, 0, 0);
        };
        constexpr int AHEAD = T::ASLOT - 1;                 // K-tiles of A in flight (1 or 2)
        if (nk > 0) { stage_W(0, kt0 * T::BKT); stage_A(0, kt0 * T::BKT); }
        if (AHEAD == 2 && nk > 1) {
            stage_A(1, (kt0 + 1) * T::BKT);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (g == 1) __builtin_amdgcn_s_barrier();           // stagger the second group by one phase
        for (int kt = 0; kt < nk; ++kt) {
            const unsigned char* sA = ringA + (kt % T::ASLOT) * T::A_BYTES;
            const unsigned char* sB = ringW + (kt & 1) * T::W_BYTES;
            const bool a_ahead = kt + AHEAD < nk;           // an A tile stays in flight across the end of this tile
            // LOAD(kt, 0)
            if (!MODCR_DBG(p.order & 32)) {                   // bit5: timing-only, no refill
                if (kt + 1 < nk) stage_W((kt + 1) & 1, (kt0 + kt + 1) * T::BKT);
                if (AHEAD == 1) { if (kt + 1 < nk) stage_A((kt + 1) & 1, (kt0 + kt + 1) * T::BKT); }
                else if (a_ahead) stage_A((kt + 2) % T::ASLOT, (kt0 + kt + 2) * T::BKT);
            }
            load_frags(sA, sB, 0, 0);
            load_frags(sA, sB, 1, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // MFMA(kt, 0)
            mfma_block();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // LOAD(kt, 1)
            load_frags(sA, sB, 0, 2);
            load_frags(sA, sB, 1, 3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (g == 1) {
                if (AHEAD == 2 && a_ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // MFMA(kt, 1)
            mfma_block();
            if (g == 0) {
                if (AHEAD == 2 && a_ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (g == 0) __builtin_amdgcn_s_barrier();           // same barrier count for both groups
    } else {
    #pragma unroll
        for (int t = 0; t < PRE; ++t)
            if (t < nk) stage(t, (kt0 + t) * T::BKT);
        for (int kt = 0; kt < nk; ++kt) {
            const int rem = nk - 1 - kt;                                // tiles issued after kt
            if (rem >= PRE - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((PRE - 1) * T::CPW) : "memory");
            else if (PRE >= 3 && rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::CPW) : "memory");
            else if (PRE >= 4 && rem == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * T::CPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // The refill of the slot freed one iteration ago is issued in pieces BETWEEN this tile's MFMA
            // groups: an LDS-DMA costs its wave ~60+ issue cycles, and with all waves of the group in
            // lock-step behind the barrier a DMA burst at the top of the tile leaves the matrix pipe idle.
            const bool refill = kt + PRE < nk;
            const int rslot = (kt + PRE) % T::NSLOT, rk0 = (kt0 + kt + PRE) * T::BKT;
            auto stage_piece = [&](int q) {
                __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[q] + rk0),
                                                 (lptr_t)(smem + rslot * T::STAGE + (wave + q * T::NWAVES) * 1024), 16, 0, 0);
            };
            const unsigned char* sA = smem + (kt % T::NSLOT) * T::STAGE;
            const unsigned char* sB = sA + BM * T::ROWB;
            // register double-buffered fragments: the reads of k-step ks+1 are in flight while the MFMAs
            // of k-step ks issue
            bf16x8 fa[2][TM], fb[2][TN];
            auto load_frags = [&](int set, int ks) {
    #pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[set][i] = *reinterpret_cast<const bf16x8*>(sA + T::off((wm * TM + i) * 32 + r, ks * 2 + h));
    #pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[set][j] = *reinterpret_cast<const bf16x8*>(sB + T::off((wn * TN + j) * 32 + r, ks * 2 + h));
            };
            constexpr int NGRP = KSTEPS * TM;                   // MFMA groups (one row of wave tiles each) per K-tile
            load_frags(0, 0);
    #pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks + 1 < KSTEPS) load_frags((ks + 1) & 1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE this k-step's MFMAs
    #pragma unroll
                for (int i = 0; i < TM; ++i) {
    #pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
                    // DMA pieces spread evenly over the NGRP groups
                    const int g = ks * TM + i;
    #pragma unroll
                    for (int q = 0; q < T::CPW; ++q)
                        if (q * NGRP / T::CPW == g && refill) stage_piece(q);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    __syncthreads();          // the ring becomes the epilogue's staging area

    // epilogue: accumulators -> LDS (fp32 [SLAB][BN], reuses the staging ring) -> coalesced
    // row-contiguous stores with bias / activation / residual applied on the way out.
    float* sC = reinterpret_cast<float*>(smem);
    constexpr int CQ = BN / 4;                      // 16-byte chunks per row
    constexpr int RPI = T::NT / CQ;                 // rows stored per iteration
    constexpr int ITERS = T::SLAB / RPI;
    const int cq = tid % CQ, rq = tid / CQ;
    const int n = n0 + cq * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) if (n + c < p.N) bv[c] = p.bias[n + c];
    }
    const bool vec = p.vec_ok && (n + 3 < p.N);
    for (int slab = 0; slab < BM / T::SLAB; ++slab) {
        if (slab) __syncthreads();      // everyone has stored the previous slab
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int trow = (wm * TM + i) * 32;    // first row of this 32-row MFMA tile
            if (trow / T::SLAB != slab) continue;   // wave-uniform
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = trow - slab * T::SLAB + (e & 3) + 8 * (e >> 2) + 4 * h;
                    sC[row * BN + (wn * TN + j) * 32 + r] = acc[i][j][e];
                }
        }
        __syncthreads();
        const int mb = m0 + slab * T::SLAB;
        if (vec) {
            // residual loads are issued a batch ahead (they are independent of LDS), then read / fuse / store
            constexpr int BATCH = ITERS < 8 ? ITERS : 8;
#pragma unroll
            for (int b0 = 0; b0 < ITERS; b0 += BATCH) {
                bf16x4 rb[BATCH];
                float4 rf[BATCH];
#pragma unroll
                for (int it = 0; it < BATCH; ++it) {
                    if (b0 + it >= ITERS) break;
                    const int m = min(mb + (b0 + it) * RPI + rq, p.M - 1);
                    if (RES == 1) rb[it] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.res) + (int64_t)m * p.ldr + n);
                    if (RES == 2) rf[it] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + (int64_t)m * p.ldr + n);
                }
#pragma unroll
                for (int it = 0; it < BATCH; ++it) {
                    if (b0 + it >= ITERS) break;
                    const int row = (b0 + it) * RPI + rq;
                    const int m = mb + row;
                    const float4 cv = *reinterpret_cast<const float4*>(sC + row * BN + cq * 4);
                    float v[4] = {cv.x, cv.y, cv.z, cv.w};
                    if (!MODCR_DBG(p.order & 16)) bias_act4(v, bv, ACT);   // bit4: timing-only, skip bias/activation
                    if (RES == 1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] += (float)rb[it][c];
                    }
                    if (RES == 2) { v[0] += rf[it].x; v[1] += rf[it].y; v[2] += rf[it].z; v[3] += rf[it].w; }
                    if (m < p.M && !MODCR_DBG(p.order & 8)) {          // bit3: timing-only, skip the stores
                        if (OUT != MODCR_F32) {
                            o16x4<OUT> o;
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] = cvt16<OUT>(v[c]);
                            *reinterpret_cast<o16x4<OUT>*>(reinterpret_cast<bf16*>(p.C) + (int64_t)m * p.ldc + n) = o;
                        } else {
                            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n) =
                                make_float4(v[0], v[1], v[2], v[3]);
                        }
                    }
                }
            }
        } else {
            for (int it = 0; it < ITERS; ++it) {
                const int row = it * RPI + rq;
                const int m = mb + row;
                if (m >= p.M) continue;
                for (int c = 0; c < 4; ++c) {
                    if (n + c >= p.N) continue;
                    float t = act_apply(sC[row * BN + cq * 4 + c] + bv[c], ACT);
                    if (RES == 1) t += (float)reinterpret_cast<const bf16*>(p.res)[(int64_t)m * p.ldr + n + c];
                    if (RES == 2) t += reinterpret_cast<const float*>(p.res)[(int64_t)m * p.ldr + n + c];
                    if (OUT != MODCR_F32) reinterpret_cast<typename Out16<OUT>::T*>(p.C)[(int64_t)m * p.ldc + n + c] = cvt16<OUT>(t);
                    else reinterpret_cast<float*>(p.C)[(int64_t)m * p.ldc + n + c] = t;
                }
            }
        }
    }
}

typedef Tile<128, 128, 2, 2, 64, 2> TileS;      // 2 x 32 KB (measured: deeper rings of 32-wide tiles are no faster)
typedef Tile<256, 256, 2, 4, 64, 2> TileL;      // 2 x 64 KB
typedef Tile<192, 256, 2, 4, 64, 2, 3> TileM;   // A ring 3 x 24 KB + W ring 2 x 32 KB, wave tile 96x64

// ---- 256 x 256 tile, 8 waves, half-tile ring ("8-phase" schedule) ---------------------------------
// The ring tiles above hand the matrix pipe a whole K-tile (64 KB) per barrier pair and then wait for
// the next one: L2->LDS DMA lands ~1-2 us after issue, a K-tile computes in < 1 us.  Here the unit of
// staging, waiting and computing is a HALF tile (128 rows x 64 k = 16 KB, two DMA instructions per
// wave): a K-tile is four half-tiles {A0, B0, B1, A1} and four phases, phase q multiplies one
// quadrant of the wave's 128 x 64 output by the full 64-wide K-tile.  Every phase issues its LDS
// reads, stages ONE half-tile six phases ahead, waits with a counted vmcnt that leaves four
// half-tiles (64 KB) in flight, and runs 16 MFMAs between two raw barriers.  The two wave groups
// (wr = 0 / 1, one wave of each per SIMD) run one barrier apart, so one group's MFMA cluster covers
// the other's LDS reads and DMA issue.
//   half-tile g = 4 T + {0:A0, 1:B0, 2:B1, 3:A1} of K-tile T lives in slot (T & 1) * 4 + (g & 3)
//   first read:  A0, B0 in phase 4T (B0 stays in registers for phase 4T+3), B1 in 4T+1, A1 in 4T+2
//   staged:      in phase g - 6, i.e. >= 2 phases after the slot's last read (the other group may be
//                one barrier behind), and waited for (vmcnt(8) after staging g+... ) in the phase
//                before its first read: wait -> barrier -> read in the NEXT phase.
// Needs N % 256 == 0, K % 128 == 0 (two K-tiles per loop trip); rows are clamped / masked.
struct P8 {
    [[maybe_unused]] static constexpr int BM = 256, BN = 256, NT = 512;
    static constexpr int HALF = 128 * 128;            // bytes per half-tile
    static constexpr int SMEM = 10 * HALF;            // 128 KB ring + 32 KB epilogue staging
};

// Persistent: workgroup b walks tiles b, b + gridDim, ... .  After a tile's last phase the NEXT tile's
// first six half-tiles are put in flight, then the finished tile's epilogue runs (through the 32 KB of
// LDS those six do not touch), so DMA latency and the store burst overlap the next K loop.  The counted
// waits of the next tile do NOT allow for the epilogue's stores: stores retire out of order with respect to
// the older LDS-DMA loads (measured: with vmcnt(8 + stores) a tile occasionally started on half-tiles that
// had not landed -- 1 launch in ~150 without a bias load to serialise them), so every wait is
// "at most N operations of any kind outstanding", which bounds the outstanding DMAs whatever the stores do.
// DIRECT = 1: the product is accumulated transposed (weights as the MFMA A operand: a lane holds four consecutive
// output columns of one row), so the epilogue needs no LDS pass: bias / activation / residual in registers, two
// column blocks exchanged between lane rows (v_permlane16_swap) into 16-byte bf16 stores, or plain 16-byte fp32 stores.
// SPEC = 1 (DIRECT, 16-bit output, no residual): the ring never drains and no counted wait ever follows a store.
//   * the last trip of a tile stages the NEXT tile's first six half-tiles in its phases 2..7 (the ring simply continues: same
//     slots, same six-phases-ahead distance, same vmcnt(8)), so there is no prologue burst between two tiles;
//   * the epilogue opens with vmcnt(0) -- only the next tile's half-tiles 2..5 are outstanding, the youngest issued a phase
//     earlier, and no store yet -- so half-tiles 0..5 of the next tile are KNOWN to have landed before the first store goes out and
//     phases 0..3 of the next K loop wait for nothing.  The first counted wait (phase 4) comes four phases after the last store.
//   (vmcnt counts loads and stores together and stores retire out of order with the LDS-DMA loads: in the plain kernel the wave
//   that has just stored waits at the next tile's entry until its own store burst has drained -- 4.6 k of a tile's 39 k cycles on
//   the FFN-up shape, tools/trace_gemm.py.)
template <int ACT, int RES, int OUT, int DIRECT, int TN = 0, int SPEC = 0>
__global__ __launch_bounds__(512, 2) void linear_bf16_p8_kernel(LinearArgs p) {
    static_assert(!SPEC || (DIRECT && TN == 0 && RES == 0 && OUT != MODCR_F32), "seamless ring: register epilogue, 16-bit output");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // split-K (dW of the backward: K = tokens): work item = (split, tile); a split covers k_tiles_per_split K-tiles
    // from its own K offset and writes its own fp32 partial at C + split * split_stride
    const int tmn = p.tiles_m * p.tiles_n;
    const int nsplit = p.k_tiles_per_split ? (p.K >> 6) / p.k_tiles_per_split : 1;
    const int nwg = tmn * nsplit;
    char* Cb = reinterpret_cast<char*>(p.C);
    int kA = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;
    // position v of the persistent walk -> tile; rev_walk: the walk runs from the LAST tile to the first (see launch_p8d: a consumer that
    // starts on the rows its producer wrote last finds them in the Infinity Cache)
    auto tile_at = [&](int v, int n) { const int t = xcd_remap(v, n); return p.rev_walk ? n - 1 - t : t; };
    // tuning build, A/B only (MI355X_MICROARCH.md "Two waves per SIMD" item 4: static priority for the younger half): order bit 8 raises
    // waves 4-7 for the whole kernel, bit 12 waves 0-3 (the control)
    if (MODCR_DBG((p.order & 256) && !(p.order & 64)) && wr == 1) __builtin_amdgcn_s_setprio(1);
    if (MODCR_DBG((p.order & 4096) && !(p.order & 64)) && wr == 0) __builtin_amdgcn_s_setprio(1);

    // DMA sources.  Half-tile = 16 pieces of 1 KiB (8 rows x 128 B), pieces wave and wave + 8.
    // LDS row r of A-half mh = X row m0 + 128 mh + r; LDS row r of B-half nh = W row
    // n0 + 64 (r / 32) + 32 nh + r % 32, so that wave column wc owns output columns n0 + 64 wc .. +63.
    // Addresses are a uniform base (SGPR pair: matrix + tile + k offset) plus a 32-bit per-lane byte
    // offset, so advancing along K costs no vector instructions and a source takes one register.
    // a wave-uniform pointer pinned to SGPRs (the per-lane part of every address below is a 32-bit offset)
    auto uniform_ptr = [](const void* q) {
        const uint64_t b64 = reinterpret_cast<uint64_t>(q);
        return reinterpret_cast<char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(b64 & 0xffffffffu)));
    };
    unsigned offAsrc[2][2], offBsrc[2];
    [[maybe_unused]] unsigned offA1 = 0, offB1 = 0;     // SPEC: tile-invariant per-lane offsets (set once), the tile lives in the uniform bases
    [[maybe_unused]] const bf16* baseA = p.A;
    const bf16* baseB;
    auto set_sources = [&](int m0, int n0, int ks) {
        if constexpr (SPEC) {
            // whole tiles only (M % 256 == 0): piece + 8 = LDS row + 64 = X row + 64 (A) / W row + 128 (B), same swizzle key; A-half
            // + 1 = X row + 128 -- all uniform strides, so a tile switch is scalar arithmetic and costs no vector register
            baseA = reinterpret_cast<const bf16*>(uniform_ptr(p.A + (int64_t)m0 * p.lda + ks));
            baseB = reinterpret_cast<const bf16*>(uniform_ptr(p.W + (int64_t)n0 * p.ldw + ks));
            return;
        }
        if constexpr (TN == 2) {
            // half-TN product (dW = dY^T X with dY^T transposed by the caller, X token-major as the layer saved it): the A side is the
            // row-major form below, the B side the token-major form of TN = 1 (its image, swizzle and transposed fragment reads)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = (wave + 8 * q) * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    offAsrc[hh][q] = (unsigned)(((int64_t)min(m0 + 128 * hh + r, p.M - 1) * p.lda + c * 8) * 2);
                const int rt = (wave + 8 * q) * 4 + (lane >> 4), pc = lane & 15;
                const int ct = pc ^ (((rt & 3) << 1) | (((rt >> 3) & 1) << 3));
                offBsrc[q] = (unsigned)(((int64_t)rt * p.ldw + n0 + 64 * (ct >> 2) + 8 * (ct & 3)) * 2);
            }
            baseB = p.W;
            kA = ks;
            return;
        }
        if constexpr (TN == 1) {
            // TN product (dW = dY^T X, both operands token-major [tokens][features]): a half-tile image is [64 tokens]
            // [128 features] = 256-byte rows; piece (wave + 8 q) = token rows 4 piece .. + 3, lane = (row, 16-byte chunk).
            // Chunks are XOR-swizzled on the SOURCE side by the row (32-byte spans by (row & 3) | ((row >> 3) & 1) << 2)
            // so that the transposed fragment reads (ds_read_b64_tr_b16: 4 rows x 16 columns per 16-lane group) of a
            // 32-lane half land on distinct banks.  The B image keeps a wave column's 64 features split 32 | 32 over
            // the halves nh: image chunk c <-> feature 64 (c / 4) + 32 nh + 8 (c % 4).
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = (wave + 8 * q) * 4 + (lane >> 4), pc = lane & 15;
                const int c = pc ^ (((r & 3) << 1) | (((r >> 3) & 1) << 3));
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
                    offAsrc[hh][q] = (unsigned)(((int64_t)r * p.lda + m0 + 128 * hh + c * 8) * 2);
                offBsrc[q] = (unsigned)(((int64_t)r * p.ldw + n0 + 64 * (c >> 2) + 8 * (c & 3)) * 2);
            }
            baseB = p.W;
            kA = ks;
            return;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wave + 8 * q) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                offAsrc[hh][q] = (unsigned)(((int64_t)min(m0 + 128 * hh + r, p.M - 1) * p.lda + c * 8) * 2);
            offBsrc[q] = (unsigned)(((int64_t)(64 * (r >> 5) + (r & 31)) * p.ldw + c * 8) * 2);
        }
        baseB = p.W + (int64_t)n0 * p.ldw + ks;
        kA = ks;
    };
    // slot order inside a K-tile buffer: A0, B0, B1, A1
    auto stage_half = [&](int buf, int kind, int k0) {
        if constexpr (SPEC) {
            const bool isA = (kind == 0 || kind == 3);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const char* base = isA ? uniform_ptr(baseA + k0 + (int64_t)((kind == 3 ? 128 : 0) + 64 * q) * p.lda)
                                       : uniform_ptr(baseB + k0 + (int64_t)((kind == 2 ? 32 : 0) + 128 * q) * p.ldw);
                glds16(isA ? offA1 : offB1, base, smem + (buf * 4 + kind) * P8::HALF + (wave + 8 * q) * 1024);
            }
            return;
        }
        // explicitly scalar, or loop strength reduction turns the sources into per-lane 64-bit pointers
        const char* base;
        if constexpr (TN == 2) {
            const int kt_ = min(kA + k0, p.kvalid - 64);              // uniform: K-tiles past the last token re-read the last valid one
            base = uniform_ptr((kind == 0 || kind == 3) ? (const void*)(p.A + kA + k0)
                                                        : (const void*)(baseB + (int64_t)kt_ * p.ldw + (kind == 2 ? 32 : 0)));
        } else {
            base = TN ? uniform_ptr((kind == 0 || kind == 3) ? (const void*)(p.A + (int64_t)(kA + k0) * p.lda)
                                                             : (const void*)(baseB + (int64_t)(kA + k0) * p.ldw + (kind == 2 ? 32 : 0)))
                      : uniform_ptr((kind == 0 || kind == 3) ? (const void*)(p.A + kA + k0)
                                                             : (const void*)(baseB + (int64_t)(kind == 2 ? 32 : 0) * p.ldw + k0));
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned off = (kind == 0) ? offAsrc[0][q] : (kind == 3) ? offAsrc[1][q] : offBsrc[q];
            __builtin_amdgcn_global_load_lds((gptr_t)(base + off),
                                             (lptr_t)(smem + (buf * 4 + kind) * P8::HALF + (wave + 8 * q) * 1024), 16, 0, 0);
        }
    };
    // half-tiles 0..5 = A0 B0 B1 A1 of K-tile 0, A0 B0 of K-tile 1 (slots 6, 7 stay free for the epilogue)
    auto prologue = [&]() {
        stage_half(0, 0, 0); stage_half(0, 1, 0); stage_half(0, 2, 0); stage_half(0, 3, 0);
        stage_half(1, 0, 64); stage_half(1, 1, 64);
    };
    if constexpr (SPEC) {
        const int r = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        offA1 = (unsigned)(((int64_t)r * p.lda + c * 8) * 2);
        offB1 = (unsigned)(((int64_t)(64 * (r >> 5) + (r & 31)) * p.ldw + c * 8) * 2);
    }

    // per-lane LDS read offsets: row block base + swizzled chunk for k-step 0 / 1
    // Eight base registers ([A|B][buffer][k-step], made opaque so they are neither re-derived from their
    // parts nor multiplied per slot); slot and row-block offsets ride in the instructions' immediates.
    typedef const __attribute__((address_space(3))) bf16x8* lds_v8;
    const int keyr = (l15 >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned ck0 = ((l4 ^ keyr) & 7) << 4, ck1 = (((l4 + 4) ^ keyr) & 7) << 4;
    unsigned aA[2][2], aB[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        aA[b][0] = lds0 + b * 4 * P8::HALF + (wr * 64 + l15) * 128 + ck0;
        aA[b][1] = lds0 + b * 4 * P8::HALF + (wr * 64 + l15) * 128 + ck1;
        aB[b][0] = lds0 + b * 4 * P8::HALF + (wc * 32 + l15) * 128 + ck0;
        aB[b][1] = lds0 + b * 4 * P8::HALF + (wc * 32 + l15) * 128 + ck1;
        asm volatile("" : "+v"(aA[b][0]), "+v"(aA[b][1]), "+v"(aB[b][0]), "+v"(aB[b][1]));
    }

    f32x4 acc[2][2][4][2];
    bf16x8 fa[4][2], fb[2][2][2];
    const int nk = p.k_tiles_per_split ? p.k_tiles_per_split : (p.K >> 6);      // K-tiles per work item (even, >= 4)
    // tile walk: row-major over (m, n), or -- ngroup > 0 -- column groups of `ngroup` tiles walked row by row one after the
    // other, so that the workgroups of an XCD (a contiguous chunk of the walk) stay on ONE weight slice that fits their L2
    auto tile_mn = [&](int t2, int& tm, int& tn) {
        if (p.ngroup) {
            const int per = p.tiles_m * p.ngroup, grp = t2 / per, r = t2 - grp * per;
            tm = r / p.ngroup;
            tn = grp * p.ngroup + (r - tm * p.ngroup);
        } else {
            tm = t2 / p.tiles_n;
            tn = t2 - tm * p.tiles_n;
        }
    };
    auto set_tile = [&](int t) {
        const int sp = t / tmn, t2 = t - sp * tmn;
        int tm, tn;
        tile_mn(t2, tm, tn);
        set_sources(tm * 256, tn * 256, sp * nk * 64);
    };
    // TN: transposed reads.  Lane (l4, l15): q4 = l15 / 4 picks the token row of the 4-row block, p4 = l15 % 4 the
    // 8-byte piece of its 32-byte span; block = tokens 8 l4 .. + 3 (second read: + 4 .. + 7) x the 16 features of
    // output block i (A) / j (B).  One base register per block and buffer (the span XOR permutes the blocks per lane).
    typedef __attribute__((address_space(3))) bf16x4* lds_tr;
    unsigned tA[2][4], tB[2][2];
    if constexpr (TN) {
        const int q4 = l15 >> 2, p4 = l15 & 3;
        const unsigned x = (unsigned)(q4 | ((l4 & 1) << 2));
        const unsigned rowoff = (unsigned)((8 * l4 + q4) * 256 + 8 * p4);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if constexpr (TN == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) tA[b][i] = lds0 + b * 4 * P8::HALF + rowoff + 32 * ((unsigned)(wr * 4 + i) ^ x);
                asm volatile("" : "+v"(tA[b][0]), "+v"(tA[b][1]), "+v"(tA[b][2]), "+v"(tA[b][3]));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) tB[b][j] = lds0 + b * 4 * P8::HALF + rowoff + 32 * ((unsigned)(wc * 2 + j) ^ x);
            asm volatile("" : "+v"(tB[b][0]), "+v"(tB[b][1]));
        }
    }
    auto tr8 = [&](unsigned addr) {                          // 8 consecutive tokens of one feature: two transposed reads
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr)(addr));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_tr)(addr + 1024));
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
        return o;
    };
    auto rdA = [&](int buf, int mh) {
        if constexpr (TN == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i][0] = tr8(tA[buf][i] + (mh ? 3 : 0) * P8::HALF);
                fa[i][1] = tr8(tA[buf][i] + (mh ? 3 : 0) * P8::HALF + 8192);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = *(lds_v8)(aA[buf][0] + (mh ? 3 : 0) * P8::HALF + i * 2048);
            fa[i][1] = *(lds_v8)(aA[buf][1] + (mh ? 3 : 0) * P8::HALF + i * 2048);
        }
    };
    auto rdB = [&](int buf, int nh) {
        if constexpr (TN) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[nh][j][0] = tr8(tB[buf][j] + (1 + nh) * P8::HALF);
                fb[nh][j][1] = tr8(tB[buf][j] + (1 + nh) * P8::HALF + 8192);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[nh][j][0] = *(lds_v8)(aB[buf][0] + (1 + nh) * P8::HALF + j * 2048);
            fb[nh][j][1] = *(lds_v8)(aB[buf][1] + (1 + nh) * P8::HALF + j * 2048);
        }
    };
    // one phase: I = phase index inside the 8-phase trip (two K-tiles).
    // MODE 0 = steady state, 1 = last trip of a tile, 2 = first trip after an epilogue.
    // SPEC: the last trip of a tile that has a successor is a steady-state trip too (same code: a separate instantiation made the
    // register allocator spill fragments inside it) whose phases 2..7 stage the successor's half-tiles 0..5: at its phase 2 the
    // uniform bases move to the successor MINUS the nk K-tiles of this tile, so the running k offset continues unchanged.
    int next_m0 = 0, next_n0 = 0;      // (worked out at the tile's top: the integer divisions of the tile walk need vector registers)
    bool more_s = false;
    auto phase = [&](auto I_, auto MODE_, int kt) {
        constexpr int I = decltype(I_)::value, MODE = decltype(MODE_)::value;
        constexpr int Q = I & 3, BUF = I >> 2;
        constexpr int MH = (Q >= 2), NH = (Q == 1 || Q == 2);
        if constexpr (Q == 0) { rdB(BUF, 0); __builtin_amdgcn_sched_barrier(0); rdA(BUF, 0); }
        if constexpr (Q == 1) rdB(BUF, 1);
        if constexpr (Q == 2) rdA(BUF, 1);
        // stage half-tile g = p + 6: kind (I + 2) & 3 of K-tile kt + (I + 6) / 4
        if constexpr (SPEC && MODE == 0 && I == 2) {
            if (kt + 2 == nk) set_sources(next_m0, next_n0, -(nk << 6));   // the current tile's last two half-tiles went out in phases 0, 1
        }
        if constexpr (MODE != 1 || I < 2) {
            constexpr int KIND = (I + 2) & 3, DT = (I + 6) >> 2;
            stage_half(DT & 1, KIND, (kt + DT) << 6);
        }
        // leave min(4, remaining) half-tiles in flight
        constexpr int FLY = MODE != 1 ? 4 : (5 - I > 4 ? 4 : (5 - I < 0 ? 0 : 5 - I));
        constexpr int VM = 2 * FLY;
        // SPEC, first trip of a tile: half-tiles 0..5 landed before the previous epilogue's stores went out (vmcnt(0) there)
        if constexpr (!(SPEC && MODE == 2 && I < 4)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if constexpr (DIRECT) acc[MH][NH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[NH][j][ks], fa[i][ks], acc[MH][NH][i][j], 0, 0, 0);
                    else acc[MH][NH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb[NH][j][ks], acc[MH][NH][i][j], 0, 0, 0);
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

    // ---- epilogue of one finished tile: per 32-row block the wave's 32 x 64 fp32 values go through its
    // private 8 KB of LDS (slots 6/7 and the 32 KB above the ring; 16-column groups XOR-swizzled by
    // (row >> 2) & 3) and come back row-contiguous: 8 lanes x 8 columns per row = full 128-byte (bf16) /
    // 256-byte (fp32) lines, bias / activation / residual applied on the way out.
    unsigned char* wbuf = smem + (wave < 4 ? 6 * P8::HALF + wave * 8192 : 8 * P8::HALF + (wave - 4) * 8192);
    const int c8 = (lane & 7) * 8;
    constexpr int OSZ = (OUT == MODCR_F32 ? 4 : 2);
    const unsigned out_lane = (unsigned)(((int64_t)(lane >> 3) * p.ldc + c8) * OSZ);       // row (lane / 8), columns c8..c8+7
    const unsigned res_lane = (unsigned)(((int64_t)(lane >> 3) * p.ldr + c8) * (RES == 1 ? 2 : 4));
    auto epilogue = [&](auto FULL_, int m0, int n0, const bf16x8 (&rb)[2][8], const float (&bias8)[8]) {
        constexpr bool FULL = decltype(FULL_)::value;
        const int gn0 = n0 + wc * 64;
        // MUL_GELU_GRAD with C2: column sums of the fp32 values this tile stores (= the bias gradient of the layer that produced the
        // GELU input: d_b1 = colsum(d_u)), so that the weight-gradient product behind it need not transpose d_u to get them
        [[maybe_unused]] float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            const int rbase = m0 + mh * 128 + wr * 64;
            f32x4 rf[8][2];
            if constexpr (RES == 2) {
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    const int gmu = min(rbase + ps * 8, p.M - 8);      // ragged tiles: M % 8 == 0 is required with a residual
                    const float* rp = reinterpret_cast<const float*>(
                        uniform_ptr(reinterpret_cast<const float*>(p.res) + (int64_t)gmu * p.ldr + gn0) + res_lane);
                    rf[ps][0] = *reinterpret_cast<const f32x4*>(rp);
                    rf[ps][1] = *reinterpret_cast<const f32x4*>(rp + 4);
                }
            }
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {          // 32-row blocks: i = 2 ib, 2 ib + 1
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int row = ii * 16 + l4 * 4 + e, cg = (nh * 2 + j) ^ l4;
                                *reinterpret_cast<float*>(wbuf + row * 256 + cg * 64 + l15 * 4) = acc[mh][nh][2 * ib + ii][j][e];
                            }
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int ps = ib * 4 + pp;
                    const int row = pp * 8 + (lane >> 3);
                    const int cg = ((lane & 7) >> 1) ^ ((row >> 2) & 3);
                    const unsigned char* src = wbuf + row * 256 + cg * 64 + (lane & 1) * 32;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 16);
                    float a4[4] = {v0[0], v0[1], v0[2], v0[3]}, b4[4] = {v1[0], v1[1], v1[2], v1[3]};
                    const float ba[4] = {bias8[0], bias8[1], bias8[2], bias8[3]}, bb[4] = {bias8[4], bias8[5], bias8[6], bias8[7]};
                    bias_act4(a4, ba, ACT);
                    bias_act4(b4, bb, ACT);
                    if constexpr (RES == 1) {
                        const float ra[4] = {(float)rb[mh][ps][0], (float)rb[mh][ps][1], (float)rb[mh][ps][2], (float)rb[mh][ps][3]};
                        const float rc[4] = {(float)rb[mh][ps][4], (float)rb[mh][ps][5], (float)rb[mh][ps][6], (float)rb[mh][ps][7]};
                        res_apply4<ACT>(a4, ra);
                        res_apply4<ACT>(b4, rc);
                    }
                    if constexpr (RES == 2) {
                        const float ra[4] = {rf[ps][0][0], rf[ps][0][1], rf[ps][0][2], rf[ps][0][3]};
                        const float rc[4] = {rf[ps][1][0], rf[ps][1][1], rf[ps][1][2], rf[ps][1][3]};
                        res_apply4<ACT>(a4, ra);
                        res_apply4<ACT>(b4, rc);
                    }
                    const int gmu = rbase + ib * 32 + pp * 8;           // first of the 8 rows this pass stores
                    const int gm = gmu + (lane >> 3);
                    if constexpr (ACT == MODCR_ACT_MUL_GELU_GRAD) {
                        if (p.C2 && (FULL || gm < p.M)) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { csum[e] += a4[e]; csum[4 + e] += b4[e]; }
                        }
                    }
                    if (MODCR_DBG(p.order & 128)) {      // timing-only: everything but the global stores
                        if (a4[0] + b4[3] == 12345.678f) reinterpret_cast<float*>(p.C)[tid] = a4[1];
                    } else if (FULL || gm < p.M) {
                        char* cp = uniform_ptr(Cb + ((int64_t)gmu * p.ldc + gn0) * OSZ) + out_lane;
                        if constexpr (OUT != MODCR_F32) {
                            o16x8<OUT> o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o[e] = cvt16<OUT>(a4[e]); o[4 + e] = cvt16<OUT>(b4[e]); }
                            *reinterpret_cast<o16x8<OUT>*>(cp) = o;
                        } else {
                            *reinterpret_cast<f32x4*>(cp) = f32x4{a4[0], a4[1], a4[2], a4[3]};
                            *reinterpret_cast<f32x4*>(cp + 16) = f32x4{b4[0], b4[1], b4[2], b4[3]};
                        }
                    }
                }
            }
        }
        if constexpr (ACT == MODCR_ACT_MUL_GELU_GRAD) {
            if (p.C2) {
                // lanes with equal (lane & 7) hold the same 8 columns for 8 different rows: sum over lane >> 3, then lanes 0..7 add
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = csum[e];
                    v += __shfl_xor(v, 8);
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    csum[e] = v;
                }
                if (lane < 8) {
                    float* cs = reinterpret_cast<float*>(p.C2) + gn0 + c8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) atomicAdd(cs + e, csum[e]);
                }
            }
        }
    };

    // ---- direct epilogue (DIRECT = 1): lane = row l15 of a 16-row block, columns 4 l4 .. 4 l4 + 3 of each 16-column block
    auto epilogue_direct = [&](auto FULL_, int m0, int n0) {
        constexpr bool FULL = decltype(FULL_)::value;
        const int gn0 = n0 + wc * 64;
        f32x4 bv[2][2];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bv[nh][j] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + gn0 + nh * 32 + j * 16 + 4 * l4) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int cswap = (l4 & 1) * 16 + (l4 >> 1) * 8;        // first column of the 8 a lane holds after the row swap
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gm = m0 + mh * 128 + wr * 64 + i * 16 + l15;
                const bool rowok = FULL || gm < p.M;
                const int gmc = FULL ? gm : min(gm, p.M - 1);
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    float v[2][4];
                    [[maybe_unused]] float u[2][4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float b4[4] = {bv[nh][j][0], bv[nh][j][1], bv[nh][j][2], bv[nh][j][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[j][e] = acc[mh][nh][i][j][e];
                        if constexpr (ACT == MODCR_ACT_GELU_KEEP) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) u[j][e] = v[j][e] + b4[e];
                        }
                        bias_act4(v[j], b4, ACT);
                        const int64_t roff = (int64_t)gmc * p.ldr + gn0 + nh * 32 + j * 16 + 4 * l4;
                        if constexpr (RES == 1) {
                            const bf16x4 r = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.res) + roff);
                            const float r4[4] = {(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
                            res_apply4<ACT>(v[j], r4);
                        }
                        if constexpr (RES == 2) {
                            const f32x4 r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + roff);
                            const float r4[4] = {r[0], r[1], r[2], r[3]};
                            res_apply4<ACT>(v[j], r4);
                        }
                    }
                    if constexpr (ACT == MODCR_ACT_GELU_KEEP && OUT == MODCR_BF16) {
                        // the GELU input, kept for the backward (same swap and 16-byte stores as the output below)
                        bf16x4 a = {(bf16)u[0][0], (bf16)u[0][1], (bf16)u[0][2], (bf16)u[0][3]};
                        bf16x4 b = {(bf16)u[1][0], (bf16)u[1][1], (bf16)u[1][2], (bf16)u[1][3]};
                        unsigned a0 = reinterpret_cast<const unsigned*>(&a)[0], a1 = reinterpret_cast<const unsigned*>(&a)[1];
                        unsigned b0 = reinterpret_cast<const unsigned*>(&b)[0], b1 = reinterpret_cast<const unsigned*>(&b)[1];
                        const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                        if (rowok) {
                            bf16* cp = reinterpret_cast<bf16*>(p.C2) + (int64_t)gm * p.ldc + gn0 + nh * 32 + cswap;
                            *reinterpret_cast<uint4*>(cp) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        }
                    }
                    if constexpr (OUT != MODCR_F32) {
                        o16x4<OUT> a = {cvt16<OUT>(v[0][0]), cvt16<OUT>(v[0][1]), cvt16<OUT>(v[0][2]), cvt16<OUT>(v[0][3])};
                        o16x4<OUT> b = {cvt16<OUT>(v[1][0]), cvt16<OUT>(v[1][1]), cvt16<OUT>(v[1][2]), cvt16<OUT>(v[1][3])};
                        unsigned a0 = reinterpret_cast<const unsigned*>(&a)[0], a1 = reinterpret_cast<const unsigned*>(&a)[1];
                        unsigned b0 = reinterpret_cast<const unsigned*>(&b)[0], b1 = reinterpret_cast<const unsigned*>(&b)[1];
                        // odd lane rows of block j = 0 <-> even lane rows of block j = 1: even rows end up with 8 consecutive
                        // columns of block 0, odd rows with 8 consecutive columns of block 1
                        const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                        if (rowok) {
                            bf16* cp = reinterpret_cast<bf16*>(Cb) + (int64_t)gm * p.ldc + gn0 + nh * 32 + cswap;
                            *reinterpret_cast<uint4*>(cp) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                        }
                    } else {
                        if (rowok) {
                            float* cp = reinterpret_cast<float*>(Cb) + (int64_t)gm * p.ldc + gn0 + nh * 32 + 4 * l4;
                            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0][0], v[0][1], v[0][2], v[0][3]};
                            *reinterpret_cast<f32x4*>(cp + 16) = f32x4{v[1][0], v[1][1], v[1][2], v[1][3]};
                        }
                    }
                }
            }
    };

    // tuning build only (MODCR_GEMM_TRACE_PTR): cycle stamps of workgroup 0's waves 0 and 4 at the seams of every tile
    [[maybe_unused]] int trace_it = 0;
    [[maybe_unused]] auto trace = [&](int ev) {
        if (MODCR_DBG(p.order & 512)) {
            if ((int)blockIdx.x == p.trace_wg && lane == 0 && (wave & 3) == 0 && trace_it < 64) {
                reinterpret_cast<unsigned long long*>(p.C2)[(wave >> 2) * 1024 + trace_it * 8 + ev] = __builtin_readcyclecounter();
                // slot 7: the 100 MHz wall clock at the tile's top -- shader cycles per 10 ns between two tiles = the clock the chip holds
                // inside this kernel (MI355X_MICROARCH.md "DVFS give-back" item 6)
                if (ev == 0) reinterpret_cast<unsigned long long*>(p.C2)[(wave >> 2) * 1024 + trace_it * 8 + 7] = __builtin_amdgcn_s_memrealtime();
            }
        }
    };

    // ---- SPEC epilogue: vmcnt(0) FIRST -- only the next tile's half-tiles 2..5 are outstanding, the youngest issued a phase ago --
    // then epilogue_direct's arithmetic with its stores interleaved (issuing the 16 stores takes ~4 k cycles per workgroup: they must
    // overlap the activation arithmetic, not follow it).  The wave's 64 bias values were put into LDS (the 32 KB above the ring,
    // which the register epilogue does not use) by one 4-byte LDS-DMA at the tile's top -- covered by this vmcnt(0), no register
    // across the K loop, no load in front of the arithmetic.
    [[maybe_unused]] auto epilogue_spec = [&](int m0, int n0) {
        const int gn0 = n0 + wc * 64;
        // per-lane indices from an opaque copy of the lane id: derived from `lane` itself, the compiler hoists this epilogue's
        // address math out of the persistent tile loop and keeps it live (or spilled) across every K loop
        int lane_e = lane;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(lane_e) : : "memory");
        trace(4);
        // ---- L2 warm-up for the NEXT tile's activation rows (round 6; TUNING BUILD ONLY, MODCR_GEMM_PF=1 -- measured, not adopted).
        // The K loop stages a half-tile six phases (~3.4 k cycles = 1.7 us) before its fragments are read, which covers an L2 hit but not a miss:
        // a tile whose 256 activation rows nobody in this XCD has touched yet waits ~900 cycles on every K-tile -- its K loop runs
        // 36-42 k cycles instead of 27 k, and the column-group walk puts most tiles on such rows: EVERY workgroup's mean K loop is
        // 31-33 k on the FFN-up shape, in a back-to-back loop and behind a LayerNorm pass that has just written the operand alike
        // (tools/trace_gemm.py with MODCR_GEMM_TRACE_WG / CHAIN=1; profiles/r06_gemm_tile_trace_{pf,chain}.txt).  With the knob the
        // workgroups that will work on the same rows next (one per column tile of the group) share the touching of that block's
        // 128-byte lines from their epilogues -- K-tile kt of every row goes to the workgroup whose next column is kt mod G: one or two
        // 4-byte LDS-DMAs per thread into 2 KB of scratch above the ring, no register written, older than every later wait -- and the K
        // loop of every tile drops to its 27.3-27.6 k floor.  But the same reads now land in the epilogue, whose stores already run at
        // the chip's write rate, and it grows by 2.4 k of the 4 k saved; at kernel level the FFN-up shape goes 401.9 -> 396.6 us alone
        // and 495.1 -> 495.0 us behind the LayerNorm pass (tools/ab_gemm_order.py VARIANTS=..., tools/ab_ffn_chain_pf.py), K = 1024
        // with four column groups 334 -> 345 us.  (Every workgroup touching the whole block -- five DMAs per thread, 328 KB per CU
        // through its L1 -- emptied the K loop just the same and cost the epilogue 9 k.)  The reading: the tile is co-limited by how
        // the memory system takes 256 workgroups reading and then writing in lock-step, not by the instruction stream.
        if (MODCR_DBG(p.pf_next) && more_s && nk <= 16) {
            const int G = p.ngroup ? p.ngroup : p.tiles_n;
            const int c = (next_n0 >> 8) % G;
            const char* pa = uniform_ptr(reinterpret_cast<const char*>(p.A) + (int64_t)(next_m0 + wave * 32) * p.lda * 2);
            const unsigned prow = (unsigned)((lane_e >> 1) * (int)p.lda * 2);
            for (int kt = 2 + c + G * (lane_e & 1); kt < nk; kt += 2 * G)
                __builtin_amdgcn_global_load_lds((gptr_t)(pa + prow + kt * 128), (lptr_t)(smem + 8 * P8::HALF + 2048 + wave * 256), 4, 0, 0);
        }
        const int l15 = lane_e & 15, l4 = lane_e >> 4;
        const int cswap = (l4 & 1) * 16 + (l4 >> 1) * 8;
        const unsigned lane_off = (unsigned)((l15 * (int)p.ldc + cswap) * 2);
        float bq[2][2][4];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) b4 = *reinterpret_cast<const f32x4*>(smem + 8 * P8::HALF + wave * 256 + (nh * 32 + j * 16 + 4 * l4) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) bq[nh][j][e] = b4[e];
            }
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // uniform base (tile, row block) + one 32-bit per-lane offset: no vector address arithmetic per store
                char* rowbase = uniform_ptr(Cb + ((int64_t)(m0 + mh * 128 + wr * 64 + i * 16) * p.ldc + gn0) * 2);
                [[maybe_unused]] char* rowbase2 = uniform_ptr(reinterpret_cast<char*>(p.C2) + ((int64_t)(m0 + mh * 128 + wr * 64 + i * 16) * p.ldc + gn0) * 2);
                [[maybe_unused]] unsigned ua0 = 0, ua1 = 0, ub0 = 0, ub1 = 0;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    float v[2][4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[j][e] = acc[mh][nh][i][j][e];
                        if constexpr (ACT == MODCR_ACT_GELU_KEEP) {
                            // the GELU input, kept for the backward: the same swap and 16-byte stores into C2 (row stride ldc)
                            float u[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) u[e] = v[j][e] + bq[nh][j][e];
                            const bf16x4 ub = {(bf16)u[0], (bf16)u[1], (bf16)u[2], (bf16)u[3]};
                            if (j == 0) { ua0 = reinterpret_cast<const unsigned*>(&ub)[0]; ua1 = reinterpret_cast<const unsigned*>(&ub)[1]; }
                            else { ub0 = reinterpret_cast<const unsigned*>(&ub)[0]; ub1 = reinterpret_cast<const unsigned*>(&ub)[1]; }
                        }
                        bias_act4(v[j], bq[nh][j], ACT);
                    }
                    if constexpr (ACT == MODCR_ACT_GELU_KEEP) {
                        const auto t0 = __builtin_amdgcn_permlane16_swap(ua0, ub0, false, false);
                        const auto t1 = __builtin_amdgcn_permlane16_swap(ua1, ub1, false, false);
                        *reinterpret_cast<uint4*>(rowbase2 + nh * 64 + lane_off) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                    }
                    o16x4<OUT> a = {cvt16<OUT>(v[0][0]), cvt16<OUT>(v[0][1]), cvt16<OUT>(v[0][2]), cvt16<OUT>(v[0][3])};
                    o16x4<OUT> b = {cvt16<OUT>(v[1][0]), cvt16<OUT>(v[1][1]), cvt16<OUT>(v[1][2]), cvt16<OUT>(v[1][3])};
                    unsigned a0 = reinterpret_cast<const unsigned*>(&a)[0], a1 = reinterpret_cast<const unsigned*>(&a)[1];
                    unsigned b0 = reinterpret_cast<const unsigned*>(&b)[0], b1 = reinterpret_cast<const unsigned*>(&b)[1];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    if (MODCR_DBG(p.order & 128)) {          // timing-only: everything but the global stores
                        if (s0[0] == 0x12345678u) reinterpret_cast<unsigned*>(p.C)[tid] = s1[0];
                    } else {
                        if (MODCR_DBG(p.stpol)) {
                            // tuning build, A/B only (MODCR_GEMM_STPOL): the cache-policy bits of the output stores -- 1 = nt, 2 = sc1,
                            // 3 = sc0 sc1, 4 = sc1 nt.  (Round 6: what rounds 2-5 called "non-temporal stores", __builtin_nontemporal_store
                            // below, compiles to four plain global_store_dword on this toolchain -- no nt bit: its effect was the store
                            // granularity.)
                            typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
                            const u32x4s vv = {s0[0], s1[0], s0[1], s1[1]};
                            const char* sb = rowbase + nh * 64;
                            if (p.stpol == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(lane_off), "v"(vv), "s"(sb) : "memory");
                            else if (p.stpol == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(lane_off), "v"(vv), "s"(sb) : "memory");
                            else if (p.stpol == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(lane_off), "v"(vv), "s"(sb) : "memory");
                            else asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt" ::"v"(lane_off), "v"(vv), "s"(sb) : "memory");
                        } else if (MODCR_DBG(p.order & 2048)) {      // A/B: four dword stores through __builtin_nontemporal_store (see above)
                            const uint4 vv = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                            __builtin_nontemporal_store(vv.x, reinterpret_cast<unsigned*>(rowbase + nh * 64 + lane_off));
                            __builtin_nontemporal_store(vv.y, reinterpret_cast<unsigned*>(rowbase + nh * 64 + lane_off) + 1);
                            __builtin_nontemporal_store(vv.z, reinterpret_cast<unsigned*>(rowbase + nh * 64 + lane_off) + 2);
                            __builtin_nontemporal_store(vv.w, reinterpret_cast<unsigned*>(rowbase + nh * 64 + lane_off) + 3);
                        } else
                        *reinterpret_cast<uint4*>(rowbase + nh * 64 + lane_off) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    }
                }
            }
        asm volatile("" ::: "memory");
    };

    int vb = blockIdx.x;
    // (the step count shares its bits with other knobs: 2 << 10 is ALSO the non-temporal-store bit and 4 << 10 the priority one -- 1088 =
    // one step and 4160 = four are clean; round 6: neither moves FFN-up alone or behind a LayerNorm pass, profiles/r06_ab_*skew.log)
    if (MODCR_DBG(p.order & 64)) {       // timing-only: skew the workgroups' start by (b / 8 % 8) x (order >> 10) x ~0.5 us
        const int steps = ((blockIdx.x >> 3) & 7) * (p.order >> 10);
        for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(16);
    }
    {
        const int tile = tile_at(vb, nwg);
        set_tile(tile);
        prologue();
    }
    for (; vb < nwg; vb += gridDim.x) {
        const int tile = tile_at(vb, nwg);
        const int sp = tile / tmn, t2 = tile - sp * tmn;
        int tm_, tn_;
        tile_mn(t2, tm_, tn_);
        const int m0 = tm_ * 256, n0 = tn_ * 256;
        Cb = reinterpret_cast<char*>(p.C) + (int64_t)sp * p.split_stride * 4;
        trace(0);
        more_s = vb + (int)gridDim.x < nwg;
        if constexpr (SPEC) {
            set_sources(m0, n0, 0);          // (scalar: the bases the previous tile's last trip left are offset by its K extent)
            if (more_s) {
                int tm2, tn2;
                tile_mn(tile_at(vb + gridDim.x, nwg) % tmn, tm2, tn2);
                next_m0 = __builtin_amdgcn_readfirstlane(tm2 * 256);
                next_n0 = __builtin_amdgcn_readfirstlane(tn2 * 256);
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // half-tiles 0, 1 landed -> barrier -> phase 0 may read them
        // (first tile: nothing follows the prologue, wait for all of it; later tiles: at most the four younger
        // half-tiles outstanding, which also drains the previous epilogue's stores)
        if constexpr (SPEC) {
            // first tile: the prologue; later tiles: every wave waited vmcnt(0) before its stores (epilogue_spec)
            if (vb == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (vb == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();       // group 1 runs one barrier behind
        __builtin_amdgcn_sched_barrier(0);
        trace(1);
        if constexpr (SPEC) {
            if (p.bias) __builtin_amdgcn_global_load_lds((gptr_t)(p.bias + n0 + wc * 64 + lane), (lptr_t)(smem + 8 * P8::HALF + wave * 256), 4, 0, 0);
        }
        trip(std::integral_constant<int, 2>{}, 0);
        if constexpr (SPEC) {
            const int kend = more_s ? nk : nk - 2;
            for (int kt = 2; kt < kend; kt += 2) trip(std::integral_constant<int, 0>{}, kt);
            if (!more_s) trip(std::integral_constant<int, 1>{}, nk - 2);
        } else {
            for (int kt = 2; kt + 2 < nk; kt += 2) trip(std::integral_constant<int, 0>{}, kt);
            trip(std::integral_constant<int, 1>{}, nk - 2);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();       // realign: every wave has finished reading the ring
        __builtin_amdgcn_sched_barrier(0);
        trace(2);
        if constexpr (SPEC) {
            epilogue_spec(m0, n0);       // (whole tiles only: launch_p8)
            trace(6);
            ++trace_it;
            continue;
        }

        if (MODCR_DBG(p.order & 16)) {     // timing-only: no epilogue (keep the accumulators alive)
            float t = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) t += acc[a][b][i][j][0] + acc[a][b][i][j][1] + acc[a][b][i][j][2] + acc[a][b][i][j][3];
            if (t == 12345.678f) reinterpret_cast<float*>(p.C)[tid] = t;
            if (vb + (int)gridDim.x < nwg) {
                const int nt = tile_at(vb + gridDim.x, nwg);
                set_tile(nt);
                prologue();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        // bf16 residual rows of this tile: loaded BEFORE the next prologue so that waiting for them does
        // not wait for the DMAs
        if constexpr (DIRECT) {
            // (the residual is read inside the epilogue, after the next tile's prologue)
            const bool more_d = vb + (int)gridDim.x < nwg;
            if (more_d) {
                const int nt = tile_at(vb + gridDim.x, nwg);
                set_tile(nt);
                prologue();
            }
            asm volatile("" ::: "memory");
            trace(3);
            if (m0 + 256 <= p.M) {
                epilogue_direct(std::true_type{}, m0, n0);
            } else {
                epilogue_direct(std::false_type{}, m0, n0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // masked rows: operation count unknown
            }
            asm volatile("" ::: "memory");
            trace(6);
            ++trace_it;
            continue;
        }
        bf16x8 rb[2][8];
        float bias8[8];
        {
            const int gn = n0 + wc * 64 + c8;
            f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
            if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(p.bias + gn); b1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4); }
#pragma unroll
            for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
        }
        if constexpr (RES == 1) {
            const int gn = n0 + wc * 64;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int ps = 0; ps < 8; ++ps) {
                    const int gmu = min(m0 + mh * 128 + wr * 64 + ps * 8, p.M - 8);    // ragged tiles: M % 8 == 0 with a residual
                    rb[mh][ps] = *reinterpret_cast<const bf16x8*>(
                        uniform_ptr(reinterpret_cast<const bf16*>(p.res) + (int64_t)gmu * p.ldr + gn) + res_lane);
                }
        }
        asm volatile("" ::: "memory");
        const bool more = vb + (int)gridDim.x < nwg;
        if (more) {
            const int nt = tile_at(vb + gridDim.x, nwg);
            set_tile(nt);
            prologue();
        }
        asm volatile("" ::: "memory");
        if (MODCR_DBG(p.order & 32)) {   // timing-only: every tile stores to tile (0, 0): no HBM write stream
            epilogue(std::true_type{}, 0, 0, rb, bias8);
        } else if (m0 + 256 <= p.M) {
            epilogue(std::true_type{}, m0, n0, rb, bias8);
        } else {
            epilogue(std::false_type{}, m0, n0, rb, bias8);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // masked rows: operation count unknown
        }
        asm volatile("" ::: "memory");
    }
}

#ifdef MODCR_TUNING
// ---- 256 x 128 tile, FOUR waves, TWO workgroups per CU ("d4") ---------------------------------------------------------
// The 256 x 256 kernel above runs its K loop at 91 % of the matrix pipe's cycles and then leaves the pipe idle for the whole
// epilogue (bias / GELU / convert / 128 KB of stores: 30 % of an FFN-up tile, VALU-bound), because both waves of a SIMD are in the
// same stage of the same tile; a second accumulator set would not fit 256 registers.  Here the two waves of a SIMD belong to two
// INDEPENDENT workgroups (80 KB of LDS each), each on its own 256 x 128 tile: while one is in its epilogue, prologue wait or a
// barrier, the other one's MFMAs have the pipe.  Per-wave tile, fragment reads, MFMA order and register epilogue are those of the
// 256 x 256 kernel (wave = 128 x 64 outputs: results are bit-equal); what changes is the staging: a K-tile is A0 | B0 | B1 | A1 with
// A halves of 128 rows (16 KB, four LDS-DMA pieces per wave) and B halves of 64 rows (8 KB, two pieces), 1.5 x the bytes per FLOP
// of the square tile, in a 72 KB ring that holds SIX consecutive half-tiles whatever their kinds:
//   half-tile g = 4 T + {0: A0, 1: B0, 2: B1, 3: A1} lies at POS[g % 12] (a trip = 12 phases = 3 K-tiles; K % 192 == 0)
//   read:    A0, B0 in phase 4T, B1 in 4T + 1, A1 in 4T + 2 (as above)
//   staged:  in phase g - 5, after that phase's barrier (every wave is past the reads of phase g - 6, the last that touch the bytes
//            it overwrites: the table below), i.e. 4-5 phases before its first read
//   waited:  at the top of phase p everything up to half-tile p + 1 has landed: at most those of p + 2 .. p + 4 outstanding
// One barrier per phase (four waves in lockstep; the other workgroup fills the gaps).  The next tile's first five half-tiles and
// its bias row are put in flight before the epilogue.  Whole tiles only: M % 256 == 0, N % 128 == 0, K % 192 == 0.
// MEASURED AND NOT ADOPTED (round 4; tools/ab_gemm_d4.py, tools/abl_gemm_d4.py, profiles/r04_ab_gemm_d4.log): bit-equal to the
// 256 x 256 kernel on every shape, 0 differing launches in the cache-flushed stress -- and slower where it was meant to win: FFN-up
// (M = 92160, N = 3072, K = 768) 481-494 us against 392-426 us.  The epilogue IS hidden (GELU costs 5-10 us instead of 15-35), and with
// nothing staged inside the K loop the kernel takes 356 us -- the matrix pipe's rate at the clock the chip holds under this load (the
// square tile's K loops alone: 349 us), i.e. what the design was after.  But the staging adds 125 us on top: 1.5 x the bytes per FLOP
// are 24 LDS-DMA instructions of 1 KB per 512 pipe cycles and CU, 47 B/clk against the ~57 B/clk that path issues, and a wave that
// cannot issue its DMA cannot issue the MFMAs behind it either (placing the DMA instructions between the MFMAs or in front of them
// makes no difference).  At K = 3072, N = 768 it equals the square tile (339 vs 337 us).  Kept in the tuning library only.
struct D4 {
    static constexpr int NT = 256, RING = 72 * 1024, BIAS = RING, SMEM = RING + 2 * 1024;
};
__device__ __forceinline__ constexpr int d4_pos(int g) {      // byte offset of half-tile g (mod 12) in the ring
    constexpr int P[12] = {0, 16, 24, 32, 48, 64, 0, 8, 24, 40, 48, 56};
    return P[g % 12] * 1024;
}
__device__ __forceinline__ constexpr int d4_ops(int g) { return ((g & 3) == 0 || (g & 3) == 3) ? 4 : 2; }   // LDS-DMA instructions per wave
__device__ __forceinline__ const char* d4_uniform(const void* q) {
    const uint64_t b64 = reinterpret_cast<uint64_t>(q);
    return reinterpret_cast<const char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((int)(b64 & 0xffffffffu)));
}

template <int ACT, int OUT>
__global__ __launch_bounds__(256, 2) void linear_bf16_d4_kernel(LinearArgs p) {
    static_assert(OUT != MODCR_F32, "d4: 16-bit output");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int nwg = p.tiles_m * p.tiles_n;
    const int nk = p.K >> 6;                                 // K-tiles (a multiple of 3)

    // DMA sources: uniform base (tile, half, piece, k) + ONE per-lane offset per operand (piece q of a half = LDS rows 32 q + r0:
    // the swizzle key (row >> 1) & 7 does not depend on q)
    unsigned offA, offB;
    {
        const int r0 = wave * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r0 >> 1) & 7);
        offA = (unsigned)(((int64_t)r0 * p.lda + c * 8) * 2);
        offB = (unsigned)(((int64_t)r0 * p.ldw + c * 8) * 2);
    }
    const bf16* baseA = p.A;
    const bf16* baseB = p.W;
    auto stage = [&](auto G_, int k0) {                      // half-tile G (its index mod 12 and kind are compile-time), K offset k0
        constexpr int G = decltype(G_)::value, KIND = G & 3;
        unsigned char* dst = smem + d4_pos(G) + wave * 1024;
        if constexpr (KIND == 0 || KIND == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                glds16(offA, d4_uniform(baseA + k0 + (int64_t)((KIND == 3 ? 128 : 0) + 32 * q) * p.lda), dst + q * 4096);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                glds16(offB, d4_uniform(baseB + k0 + (int64_t)((KIND == 2 ? 32 : 0) + 64 * q) * p.ldw), dst + q * 4096);
        }
    };

    // LDS fragment reads: one base register per operand and k-step, the half-tile position rides in the immediate
    typedef const __attribute__((address_space(3))) bf16x8* lds_v8;
    const int keyr = (l15 >> 1) & 7;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned aA[2], aB[2];
    aA[0] = lds0 + (wr * 64 + l15) * 128 + (((l4 ^ keyr) & 7) << 4);
    aA[1] = lds0 + (wr * 64 + l15) * 128 + ((((l4 + 4) ^ keyr) & 7) << 4);
    aB[0] = lds0 + (wc * 32 + l15) * 128 + (((l4 ^ keyr) & 7) << 4);
    aB[1] = lds0 + (wc * 32 + l15) * 128 + ((((l4 + 4) ^ keyr) & 7) << 4);
    asm volatile("" : "+v"(aA[0]), "+v"(aA[1]), "+v"(aB[0]), "+v"(aB[1]));

    f32x4 acc[2][2][4][2];
    bf16x8 fa[4][2], fb[2][2][2];

    auto tile_mn = [&](int t2, int& tm, int& tn) {
        if (p.ngroup) {
            const int per = p.tiles_m * p.ngroup, grp = t2 / per, r = t2 - grp * per;
            tm = r / p.ngroup;
            tn = grp * p.ngroup + (r - tm * p.ngroup);
        } else {
            tm = t2 / p.tiles_n;
            tn = t2 - tm * p.tiles_n;
        }
    };
    auto open_tile = [&](int vb, int par) {                  // sources, bias row and half-tiles 0..4 of the workgroup's tile `vb`
        int tm, tn;
        tile_mn(xcd_remap(vb, nwg), tm, tn);
        const int m0 = __builtin_amdgcn_readfirstlane(tm * 256), n0 = __builtin_amdgcn_readfirstlane(tn * 128);
        baseA = reinterpret_cast<const bf16*>(d4_uniform(p.A + (int64_t)m0 * p.lda));
        baseB = reinterpret_cast<const bf16*>(d4_uniform(p.W + (int64_t)n0 * p.ldw));
        if (p.bias) __builtin_amdgcn_global_load_lds((gptr_t)(p.bias + n0 + wc * 64 + lane), (lptr_t)(smem + D4::BIAS + par * 1024 + wave * 256), 4, 0, 0);
        stage(std::integral_constant<int, 0>{}, 0); stage(std::integral_constant<int, 1>{}, 0);
        stage(std::integral_constant<int, 2>{}, 0); stage(std::integral_constant<int, 3>{}, 0);
        stage(std::integral_constant<int, 4>{}, 64);
    };

    // one phase: I = phase index inside the 12-phase trip (K-tiles kt .. kt + 2); LAST = the tile's last trip (nothing staged beyond it)
    auto phase = [&](auto I_, auto LAST_, int kt) {
        constexpr int I = decltype(I_)::value;
        constexpr bool LAST = decltype(LAST_)::value;
        constexpr int Q = I & 3, MH = (Q >= 2), NH = (Q == 1 || Q == 2);
        // landed: half-tiles <= I + 1 (Q == 3 reads nothing new: <= I); outstanding at most I + 2 .. I + 4 (those that exist)
        constexpr int LO = (Q == 3) ? I + 1 : I + 2;
        constexpr int HI = LAST ? (I + 4 < 11 ? I + 4 : 11) : I + 4;
        constexpr int VM = (LO <= HI ? d4_ops(LO) : 0) + (LO + 1 <= HI ? d4_ops(LO + 1) : 0) + (LO + 2 <= HI ? d4_ops(LO + 2) : 0) +
                           (LO + 3 <= HI ? d4_ops(LO + 3) : 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (Q == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[0][j][0] = *(lds_v8)(aB[0] + d4_pos(I + 1) + j * 2048);
                fb[0][j][1] = *(lds_v8)(aB[1] + d4_pos(I + 1) + j * 2048);
            }
        }
        if constexpr (Q == 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fb[1][j][0] = *(lds_v8)(aB[0] + d4_pos(I + 1) + j * 2048);
                fb[1][j][1] = *(lds_v8)(aB[1] + d4_pos(I + 1) + j * 2048);
            }
        }
        if constexpr (Q == 0 || Q == 2) {
            constexpr int GA = (Q == 0) ? I : I + 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i][0] = *(lds_v8)(aA[0] + d4_pos(GA) + i * 2048);
                fa[i][1] = *(lds_v8)(aA[1] + d4_pos(GA) + i * 2048);
            }
        }
        // the staging instructions go out between the MFMAs (the matrix pipe is busy 16 cycles per MFMA, the wave issues in 4)
        const bool early = MODCR_DBG(p.order & 2);           // A/B: stage in front of the MFMAs
        const bool nodma = MODCR_DBG(p.order & 1);           // timing-only: nothing staged inside the K loop
        if constexpr (!LAST || I + 5 < 12) { if (early && !nodma) stage(std::integral_constant<int, (I + 5) % 12>{}, (kt + ((I + 5) >> 2)) << 6); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[MH][NH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[NH][j][ks], fa[i][ks], acc[MH][NH][i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) {
                if constexpr (!LAST || I + 5 < 12) { if (!early && !nodma) stage(std::integral_constant<int, (I + 5) % 12>{}, (kt + ((I + 5) >> 2)) << 6); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto trip = [&](auto LAST_, int kt) {
        phase(std::integral_constant<int, 0>{}, LAST_, kt); phase(std::integral_constant<int, 1>{}, LAST_, kt);
        phase(std::integral_constant<int, 2>{}, LAST_, kt); phase(std::integral_constant<int, 3>{}, LAST_, kt);
        phase(std::integral_constant<int, 4>{}, LAST_, kt); phase(std::integral_constant<int, 5>{}, LAST_, kt);
        phase(std::integral_constant<int, 6>{}, LAST_, kt); phase(std::integral_constant<int, 7>{}, LAST_, kt);
        phase(std::integral_constant<int, 8>{}, LAST_, kt); phase(std::integral_constant<int, 9>{}, LAST_, kt);
        phase(std::integral_constant<int, 10>{}, LAST_, kt); phase(std::integral_constant<int, 11>{}, LAST_, kt);
    };

    // register epilogue (the 256 x 256 kernel's: lane = row l15 of a 16-row block, columns 4 l4 .. + 3 of each 16-column block; two column
    // blocks exchanged between lane rows into 16-byte stores), bias from LDS
    auto epilogue = [&](int m0, int n0, int par) {
        const int gn0 = n0 + wc * 64;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));                     // (address math not hoisted out of the tile loop)
        const int e15 = lane_e & 15, e4 = lane_e >> 4;
        const int cswap = (e4 & 1) * 16 + (e4 >> 1) * 8;
        const unsigned lane_off = (unsigned)((e15 * (int)p.ldc + cswap) * 2);
        float bq[2][2][4];
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) b4 = *reinterpret_cast<const f32x4*>(smem + D4::BIAS + par * 1024 + wave * 256 + (nh * 32 + j * 16 + 4 * e4) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) bq[nh][j][e] = b4[e];
            }
        char* Cb = reinterpret_cast<char*>(p.C);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                char* rowbase = const_cast<char*>(d4_uniform(Cb + ((int64_t)(m0 + mh * 128 + wr * 64 + i * 16) * p.ldc + gn0) * 2));
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    float v[2][4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[j][e] = acc[mh][nh][i][j][e];
                        bias_act4(v[j], bq[nh][j], ACT);
                    }
                    o16x4<OUT> a = {cvt16<OUT>(v[0][0]), cvt16<OUT>(v[0][1]), cvt16<OUT>(v[0][2]), cvt16<OUT>(v[0][3])};
                    o16x4<OUT> b = {cvt16<OUT>(v[1][0]), cvt16<OUT>(v[1][1]), cvt16<OUT>(v[1][2]), cvt16<OUT>(v[1][3])};
                    unsigned a0 = reinterpret_cast<const unsigned*>(&a)[0], a1 = reinterpret_cast<const unsigned*>(&a)[1];
                    unsigned b0 = reinterpret_cast<const unsigned*>(&b)[0], b1 = reinterpret_cast<const unsigned*>(&b)[1];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                    *reinterpret_cast<uint4*>(rowbase + nh * 64 + lane_off) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        asm volatile("" ::: "memory");
    };

    int vb = blockIdx.x, par = 0;
    if (vb < nwg) open_tile(vb, 0);
    for (; vb < nwg; vb += gridDim.x, par ^= 1) {
        int tm_, tn_;
        tile_mn(xcd_remap(vb, nwg), tm_, tn_);
        const int m0 = tm_ * 256, n0 = tn_ * 128;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma nounroll
        for (int kt = 0; kt + 3 < nk; kt += 3) trip(std::false_type{}, kt);
        trip(std::true_type{}, nk - 3);
        __builtin_amdgcn_s_barrier();                        // every wave has finished reading the ring
        __builtin_amdgcn_sched_barrier(0);
        if (vb + (int)gridDim.x < nwg) open_tile(vb + gridDim.x, par ^ 1);
        asm volatile("" ::: "memory");
        epilogue(m0, n0, par);
    }
}

#endif  // MODCR_TUNING (d4)

// ---- 192 x 384 tile, 8 waves, half-tile ring: the engine of the fused attention's QKV phase (attn.hip,
// qkv_attn4_kernel) as a plain GEMM.  Same 8-phase schedule as the 256 x 256 kernel above with half-tiles
// A = 96 rows (12 KB: 1.5 DMA pieces per wave, the half piece by lanes 0..31) and B = 192 rows (24 KB), wave
// tile 96 x 96 (2 x 4 waves), 18 MFMAs per phase, 10 DMA instructions per wave in flight.  Why a second
// shape: M = 46080 rows of the encoder give 240 x 2 tiles at N = 768 (1.9 rounds of 256 CUs, 94 % of whole
// rounds) and 240 x 8 at N = 3072 (7.5 rounds), where 256 x 256 tiles give 2.1 and 8.4; the arithmetic
// intensity is the same 128 FLOP per staged byte.  Persistent, next tile's first six half-tiles in flight
// under the epilogue (staged through the 52 KB of LDS those six do not touch).
// Needs N % 384 == 0, K % 128 == 0, K >= 256.
struct T192 {
    [[maybe_unused]] static constexpr int BM = 192, BN = 384, NT = 512;
    static constexpr int HA = 96 * 128, HB = 192 * 128;
    static constexpr int KT = 2 * HA + 2 * HB;          // A0 | A1 | B0 | B1 = 72 KB
    static constexpr int RING = 2 * KT;
    static constexpr int WB = 16 * 96 * 4;              // epilogue staging per wave: 16 rows x 96 fp32
    static constexpr int SMEM = RING + 16384;
};

// DIRECT: register epilogue as in linear_bf16_p8_kernel (transposed accumulation, no LDS pass).
template <int ACT, int RES, int OUT, int DIRECT>
__global__ __launch_bounds__(512, 2) void linear_bf16_t192_kernel(LinearArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HA = T192::HA, HB = T192::HB, KT = T192::KT;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K >> 6;
    auto tile_at = [&](int v, int n) { const int t = xcd_remap(v, n); return p.rev_walk ? n - 1 - t : t; };      // (see the p8 kernel)
    if (MODCR_DBG(p.order & 256) && wr == 1) __builtin_amdgcn_s_setprio(1);       // tuning build, A/B only: static priority (see the p8 kernel)
    if (MODCR_DBG(p.order & 4096) && wr == 0) __builtin_amdgcn_s_setprio(1);

    auto uniform_ptr = [](const void* q) {
        const uint64_t b64 = reinterpret_cast<uint64_t>(q);
        return reinterpret_cast<const char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                                             (unsigned)__builtin_amdgcn_readfirstlane((int)(b64 & 0xffffffffu)));
    };
    // DMA sources: A pieces = uniform matrix base + 32-bit per-lane offset (rows clamped to M - 1);
    // B pieces = 8 consecutive weight rows: per-piece scalar row + one per-lane offset (row lane / 8 and the
    // swizzled chunk, whose key depends only on the piece's parity = wave & 1).
    unsigned offA[2][2], vB;
    unsigned aA[2][2], aB[2][2];
    int wbrow[2][3];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int r = 8 * (wave + 8 * q);
            wbrow[nh][q] = (r / 48) * 96 + nh * 48 + r % 48;
        }
    const bf16* wt = p.W;                                   // W from row n0 on (per tile)
    // kind: 0 = A0, 1 = B0, 2 = B1, 3 = A1 (staging order); LDS order inside a K-tile buffer: A0 A1 B0 B1
    auto stage_half = [&](int buf, int kind, int k0) {
        if (kind == 0 || kind == 3) {
            const int mh = kind == 3;
            unsigned char* dst = smem + buf * KT + mh * HA;
            const char* base = uniform_ptr(p.A + k0);
            glds16(offA[mh][0], base, dst + wave * 1024);
            if ((threadIdx.x & 63) < 32) glds16(offA[mh][1], base, dst + 8192 + wave * 512);
        } else {
            const int nh = kind == 2;
            unsigned char* dst = smem + buf * KT + 2 * HA + nh * HB;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                glds16(vB, uniform_ptr(wt + (int64_t)wbrow[nh][q] * p.ldw + k0), dst + (wave + 8 * q) * 1024);
        }
    };
    // Both recomputed per tile from an opaque copy of the thread id: kept loop-invariant across tiles the address
    // registers would stay live through the epilogue (and get spilled around the K loop).
    auto set_sources = [&](int m0, int n0) {                // DMA sources of a tile (before its prologue)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int lane = tq & 63;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = q == 0 ? 8 * wave + (lane >> 3) : 64 + 4 * wave + ((lane & 31) >> 3);
                const int row = min(m0 + (r / 48) * 96 + mh * 48 + (r % 48), p.M - 1);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                offA[mh][q] = (unsigned)(((int64_t)row * p.lda + c * 8) * 2);
            }
        vB = (unsigned)((((int64_t)(lane >> 3) * p.ldw) + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 8)) * 2);
        wt = p.W + (int64_t)n0 * p.ldw;
    };
    auto set_lds_addrs = [&]() {                            // fragment read addresses (top of a tile's K loop)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int lane = tq & 63, l15 = lane & 15, l4 = lane >> 4;
        const int keyr = (l15 >> 1) & 7;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
        const unsigned ck0 = ((l4 ^ keyr) & 7) << 4, ck1 = (((l4 + 4) ^ keyr) & 7) << 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            aA[b][0] = lds0 + b * KT + (wr * 48 + l15) * 128 + ck0;
            aA[b][1] = lds0 + b * KT + (wr * 48 + l15) * 128 + ck1;
            aB[b][0] = lds0 + b * KT + 2 * HA + (wc * 48 + l15) * 128 + ck0;
            aB[b][1] = lds0 + b * KT + 2 * HA + (wc * 48 + l15) * 128 + ck1;
            asm volatile("" : "+v"(aA[b][0]), "+v"(aA[b][1]), "+v"(aB[b][0]), "+v"(aB[b][1]));
        }
    };
    // half-tiles 0..5 = A0 B0 B1 A1 of K-tile 0, A0 B0 of K-tile 1
    auto prologue = [&]() {
        stage_half(0, 0, 0); stage_half(0, 1, 0); stage_half(0, 2, 0); stage_half(0, 3, 0);
        stage_half(1, 0, 64); stage_half(1, 1, 64);
    };

    typedef const __attribute__((address_space(3))) bf16x8* lds_v8;
    f32x4 acc[2][2][3][3];
    bf16x8 fa[3][2], fb[2][3][2];
    // Only buffer 0's four read bases are kept in registers; buffer 1's are re-derived at each use (an
    // opaque add, or the compiler hoists them back into four more loop-long registers and spills).
    auto base_of = [&](unsigned b0, int buf) {
        unsigned t = b0;
        if (buf) asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(b0), "s"(KT));
        return t;
    };
    auto rdA = [&](int buf, int mh) {
        const unsigned b0 = base_of(aA[0][0], buf), b1 = base_of(aA[0][1], buf);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fa[i][0] = *(lds_v8)(b0 + mh * HA + i * 2048);
            fa[i][1] = *(lds_v8)(b1 + mh * HA + i * 2048);
        }
    };
    auto rdB = [&](int buf, int nh) {
        const unsigned b0 = base_of(aB[0][0], buf), b1 = base_of(aB[0][1], buf);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            fb[nh][j][0] = *(lds_v8)(b0 + nh * HB + j * 2048);
            fb[nh][j][1] = *(lds_v8)(b1 + nh * HB + j * 2048);
        }
    };
    // phase I of an 8-phase trip (two K-tiles).  MODE 0 = steady state, 1 = last trip of a tile.  Every wait is
    // "at most VM operations outstanding" -- the previous epilogue's stores are NOT allowed on top (they retire out
    // of order with respect to the older DMA loads, see linear_bf16_p8_kernel).
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
        constexpr int VM = MODE == 1 ? (I <= 1 ? 10 : I == 2 ? 8 : I == 3 ? 5 : I == 4 ? 2 : 0) : 10;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if constexpr (DIRECT) acc[MH][NH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[NH][j][ks], fa[i][ks], acc[MH][NH][i][j], 0, 0, 0);
                    else acc[MH][NH][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb[NH][j][ks], acc[MH][NH][i][j], 0, 0, 0);
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

    // epilogue staging: 6 KB per wave outside the six half-tiles of a prologue (A1 and B1 of buffer 1, and
    // the 16 KB above the ring)
    unsigned char* wbuf = smem + (wave < 2 ? KT + HA + wave * T192::WB
                                           : wave < 6 ? KT + 2 * HA + HB + (wave - 2) * T192::WB
                                                      : T192::RING + (wave - 6) * T192::WB);
    constexpr int OSZ = (OUT == MODCR_F32 ? 4 : 2);
    [[maybe_unused]] constexpr int RSZ = (RES == 1 ? 2 : 4);

    int vb = blockIdx.x;
    {
        const int tile = tile_at(vb, ntiles);
        set_sources((tile / p.tiles_n) * 192, (tile % p.tiles_n) * 384);
        prologue();
    }
    for (; vb < ntiles; vb += gridDim.x) {
        const int tile = tile_at(vb, ntiles);
        const int m0 = (tile / p.tiles_n) * 192, n0 = (tile % p.tiles_n) * 384;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        set_sources(m0, n0);            // again (the prologue's copies died with the previous epilogue)
        set_lds_addrs();
        // A0, B0 of K-tile 0 landed: at most the four younger half-tiles (10 instructions) outstanding, which
        // after an epilogue also drains its stores
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind
        __builtin_amdgcn_sched_barrier(0);
#pragma nounroll
        for (int kt = 0; kt + 2 < nk; kt += 2) trip(std::integral_constant<int, 0>{}, kt);
        trip(std::integral_constant<int, 1>{}, nk - 2);
        if (wr == 0) __builtin_amdgcn_s_barrier();          // realign: every wave is done with the ring
        __builtin_amdgcn_sched_barrier(0);

        if constexpr (DIRECT) {
            // ---- register epilogue: lane = row l15 of a 16-row block, columns 4 l4 .. 4 l4 + 3 of each 16-column block
            const bool more_d = vb + (int)gridDim.x < ntiles;
            const bool full_d = m0 + 192 <= p.M;
            if (more_d) {
                const int nt = tile_at(vb + gridDim.x, ntiles);
                set_sources((nt / p.tiles_n) * 192, (nt % p.tiles_n) * 384);
                prologue();
            }
            asm volatile("" ::: "memory");
            int tq = tid;
            asm volatile("" : "+v"(tq));
            const int lane = tq & 63, l15 = lane & 15, l4 = lane >> 4;
            const int gn0 = n0 + wc * 96;
            f32x4 bv[6];
#pragma unroll
            for (int b = 0; b < 6; ++b)
                bv[b] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + gn0 + b * 16 + 4 * l4) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int cswap = (l4 & 1) * 16 + (l4 >> 1) * 8;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int gm = m0 + wr * 96 + mh * 48 + i * 16 + l15;
                    const bool rowok = full_d || gm < p.M;
                    const int gmc = min(gm, p.M - 1);
                    float v[6][4];
#pragma unroll
                    for (int b = 0; b < 6; ++b) {           // column block b = 3 nh + j: columns 16 b + 4 l4 ..
                        const float b4[4] = {bv[b][0], bv[b][1], bv[b][2], bv[b][3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[b][e] = acc[mh][b / 3][i][b % 3][e];
                        bias_act4(v[b], b4, ACT);
                        const int64_t roff = (int64_t)gmc * p.ldr + gn0 + b * 16 + 4 * l4;
                        if constexpr (RES == 1) {
                            const bf16x4 r = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(p.res) + roff);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[b][e] += (float)r[e];
                        }
                        if constexpr (RES == 2) {
                            const f32x4 r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + roff);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[b][e] += r[e];
                        }
                    }
                    if constexpr (OUT != MODCR_F32) {
#pragma unroll
                        for (int pr = 0; pr < 3; ++pr) {    // blocks 2 pr, 2 pr + 1: odd lane rows of the first <-> even lane rows of the second
                            o16x4<OUT> a = {cvt16<OUT>(v[2 * pr][0]), cvt16<OUT>(v[2 * pr][1]), cvt16<OUT>(v[2 * pr][2]), cvt16<OUT>(v[2 * pr][3])};
                            o16x4<OUT> b = {cvt16<OUT>(v[2 * pr + 1][0]), cvt16<OUT>(v[2 * pr + 1][1]), cvt16<OUT>(v[2 * pr + 1][2]), cvt16<OUT>(v[2 * pr + 1][3])};
                            const unsigned a0 = reinterpret_cast<const unsigned*>(&a)[0], a1 = reinterpret_cast<const unsigned*>(&a)[1];
                            const unsigned b0 = reinterpret_cast<const unsigned*>(&b)[0], b1 = reinterpret_cast<const unsigned*>(&b)[1];
                            const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
                            if (rowok) {
                                bf16* cp = reinterpret_cast<bf16*>(p.C) + (int64_t)gm * p.ldc + gn0 + pr * 32 + cswap;
                                *reinterpret_cast<uint4*>(cp) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                            }
                        }
                    } else {
                        if (rowok) {
                            float* cp = reinterpret_cast<float*>(p.C) + (int64_t)gm * p.ldc + gn0 + 4 * l4;
#pragma unroll
                            for (int b = 0; b < 6; ++b) *reinterpret_cast<f32x4*>(cp + b * 16) = f32x4{v[b][0], v[b][1], v[b][2], v[b][3]};
                        }
                    }
                }
            if (!full_d) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // masked rows: operation count unknown
            asm volatile("" ::: "memory");
            continue;
        }
        // ---- epilogue ---------------------------------------------------------------------------------
        // per-tile opaque lane indices (see set_sources)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        const int lane = tq & 63, l15 = lane & 15, l4 = lane >> 4;
        const int gn0 = n0 + wc * 96;
        // item (it, lane): row (it * 64 + lane) / 12 of a 16-row block, columns 8 * ((it * 64 + lane) % 12) .. + 7
        int irow[3], icol[3];
#pragma unroll
        for (int it = 0; it < 3; ++it) { const int item = it * 64 + lane; irow[it] = item / 12; icol[it] = (item % 12) * 8; }
        // bf16 residual rows of this tile: loaded BEFORE the next prologue so that waiting for them does
        // not wait for the DMAs
        // bf16 residual rows: the first 48-row half is loaded BEFORE the next prologue (waiting for it does not
        // wait for the DMAs), the second half after the first half's stores
        bf16x8 rb[3][3];
        auto load_res = [&](int mh) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int gm = min(m0 + wr * 96 + mh * 48 + i * 16 + irow[it], p.M - 1);
                    rb[i][it] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16*>(p.res) + (int64_t)gm * p.ldr + gn0 + icol[it]);
                }
        };
        if constexpr (RES == 1) load_res(0);
        asm volatile("" ::: "memory");
        const bool more = vb + (int)gridDim.x < ntiles;
        const bool full = m0 + 192 <= p.M;
        if (more) {
            const int nt = tile_at(vb + gridDim.x, ntiles);
            set_sources((nt / p.tiles_n) * 192, (nt % p.tiles_n) * 384);
            prologue();
        }
        asm volatile("" ::: "memory");
        float* wf = reinterpret_cast<float*>(wbuf);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            if constexpr (RES == 1) { if (mh == 1) load_res(1); }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            wf[(4 * l4 + e) * 96 + nh * 48 + j * 16 + l15] = acc[mh][nh][i][j][e];
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const float* src = wf + irow[it] * 96 + icol[it];
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
                    float a4[4] = {v0[0], v0[1], v0[2], v0[3]}, b4[4] = {v1[0], v1[1], v1[2], v1[3]};
                    float ba[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) {
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + gn0 + icol[it]);
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + gn0 + icol[it] + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { ba[e] = b0[e]; bb[e] = b1[e]; }
                    }
                    bias_act4(a4, ba, ACT);
                    bias_act4(b4, bb, ACT);
                    const int gm = m0 + wr * 96 + mh * 48 + i * 16 + irow[it];
                    if constexpr (RES == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a4[e] += (float)rb[i][it][e]; b4[e] += (float)rb[i][it][4 + e]; }
                    }
                    if constexpr (RES == 2) {
                        const float* rp = reinterpret_cast<const float*>(p.res) + (int64_t)min(gm, p.M - 1) * p.ldr + gn0 + icol[it];
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a4[e] += r0[e]; b4[e] += r1[e]; }
                    }
                    if (full || gm < p.M) {
                        char* cp = reinterpret_cast<char*>(p.C) + ((int64_t)gm * p.ldc + gn0 + icol[it]) * OSZ;
                        if constexpr (OUT != MODCR_F32) {
                            o16x8<OUT> o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { o[e] = cvt16<OUT>(a4[e]); o[4 + e] = cvt16<OUT>(b4[e]); }
                            *reinterpret_cast<o16x8<OUT>*>(cp) = o;
                        } else {
                            *reinterpret_cast<f32x4*>(cp) = f32x4{a4[0], a4[1], a4[2], a4[3]};
                            *reinterpret_cast<f32x4*>(cp + 16) = f32x4{b4[0], b4[1], b4[2], b4[3]};
                        }
                    }
                }
            }
        }
        if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // masked rows: operation count unknown
        asm volatile("" ::: "memory");
    }
}

// Reversed tile walk (round 6).  An activation operand larger than the 256 MB Infinity Cache -- the FFN intermediate, 566 MB at 128 examples --
// was written front to back by its producer, so what the cache still holds when the consumer starts is its TAIL; a consumer that also
// walks front to back pushes that tail out with the rows it fetches first and reads every byte from HBM, one that starts at the tail reads
// the cached part from the cache.  (An operand that fits is found there in either direction.)  Tuning build: MODCR_GEMM_REV = 0 / 1 forces it.
inline int rev_walk_for(const LinearArgs& p) {
    const int knob = modcr_knob_int("MODCR_GEMM_REV", -1);
    if (knob >= 0) return knob;
    // Same-process A/B on the chain LayerNorm pass -> FFN-up -> FFN-down (tools/ab_ffn_chain_pf.py, profiles/r06_ab_ffn_chain_rev.log):
    // 817.3 -> 808.1 us at M = 92 160, 1261.9 -> 1253.1 us at M = 143 872, no change at M = 51 712 (its 318 MB mostly stay cached either
    // way); reversing the producer as well gives the gain back, as it must.  Per-tile arithmetic is untouched: outputs are bit-identical.
    return (int64_t)p.M * p.K * 2 > ((int64_t)256 << 20);
}

template <int ACT, int RES, int OUT, int DIRECT>
int launch_t192d(LinearArgs p, hipStream_t st) {
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bf16_t192_kernel<ACT, RES, OUT, DIRECT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T192::SMEM);
        if (e != hipSuccess) {
            modcr_set_error("linear: cannot reserve %d bytes of LDS: %s", T192::SMEM, hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    p.tiles_m = (p.M + 191) / 192;
    p.tiles_n = p.N / 384;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int ncu = modcr_device_cus();
    const int grid = ntiles <= ncu ? ntiles : (ncu & ~7);
    p.rev_walk = rev_walk_for(p);
    hipLaunchKernelGGL((linear_bf16_t192_kernel<ACT, RES, OUT, DIRECT>), dim3(grid), dim3(512), T192::SMEM, st, p);
    return modcr_check_launch("linear_bf16_t192");
}
template <int ACT, int RES, int OUT>
int launch_t192(const LinearArgs& p, hipStream_t st) {
    const int direct = modcr_knob_int("MODCR_GEMM_DIRECT", 1);            // tuning build only
    // with a residual the register epilogue loses (86 -> 100 us at M = 46080, N = K = 768, fp32 out): its residual
    // loads sit behind the next tile's prologue DMAs in the in-order vmcnt queue; direct == 2 forces it anyway
    return ((direct && RES == 0) || direct == 2) ? launch_t192d<ACT, RES, OUT, 1>(p, st) : launch_t192d<ACT, RES, OUT, 0>(p, st);
}
// shapes the 192 x 384 kernel takes
bool t192_ok(const LinearArgs& p) {
    if (p.k_tiles_per_split) return false;
    if (p.M < 192 || (p.N % 384) != 0 || (p.K % 128) != 0 || p.K < 256) return false;
    if ((int64_t)p.M * p.lda >= (1ll << 30) || (int64_t)384 * p.ldw >= (1ll << 30)) return false;   // 32-bit byte offsets
    if ((p.ldc % 8) != 0 || !modcr_aligned16(p.C)) return false;
    if (p.res && ((p.ldr % 8) != 0 || !modcr_aligned16(p.res))) return false;
    if (p.bias && !modcr_aligned16(p.bias)) return false;
    return true;
}

template <int ACT, int RES, int OUT, int DIRECT, int TN = 0, int SPEC = 0>
int launch_p8d(LinearArgs p, hipStream_t st) {
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bf16_p8_kernel<ACT, RES, OUT, DIRECT, TN, SPEC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, P8::SMEM);
        if (e != hipSuccess) {
            modcr_set_error("linear: cannot reserve %d bytes of LDS: %s", P8::SMEM, hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    if constexpr (TN == 2) {
        // the half-TN staging replaces a K-tile past kvalid by the last valid one: exact only for whole K-tiles of tokens (and an A operand
        // zero-padded past kvalid, which modcr_linear_bwd_weight's transposes guarantee)
        MODCR_REQUIRE((p.kvalid % 64) == 0 && p.kvalid >= 64 && p.kvalid <= p.K, "linear (half-TN): kvalid = %d must be a multiple of 64 in [64, K = %d]", p.kvalid, p.K);
    }
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = p.N / 256;
    p.order = modcr_knob_int("MODCR_GEMM_ORDER", 0);                      // tuning build only
#ifdef MODCR_TUNING
    if (getenv("MODCR_GEMM_TRACE_PTR") && !p.C2) { p.C2 = reinterpret_cast<void*>(strtoull(getenv("MODCR_GEMM_TRACE_PTR"), nullptr, 0)); p.order |= 512; }
    p.trace_wg = modcr_knob_int("MODCR_GEMM_TRACE_WG", 0);
#endif
    p.pf_next = modcr_knob_int("MODCR_GEMM_PF", 0);                      // tuning build only: A/B of the epilogue's L2 warm-up (off)
    p.stpol = modcr_knob_int("MODCR_GEMM_STPOL", 0);                     // tuning build only
    p.rev_walk = (p.k_tiles_per_split || TN) ? 0 : rev_walk_for(p);
    // Column groups (FFN-up: N = 3072, K = 768, 17 rounds of tiles per workgroup).  Walked row-major, the 32 workgroups of an XCD
    // work on 2.7 tile rows x all 12 column tiles at a time: 4.7 MB of weights against 4 MB of L2, re-fetched every round
    // (profiles/r03_gemm_pmc_ffn_up.txt: 2 x FETCH_SIZE = 1.06 GB against 146 MB of operands).  Groups of tiles_n / 2 (or / 4)
    // columns whose weight slice is <= 2.5 MB keep that slice resident: the activations are then read once per group.
    p.ngroup = 0;
    if (!p.k_tiles_per_split && p.tiles_m >= 64) {
        const int64_t wbytes = (int64_t)p.N * p.K * 2;
        for (int parts = 2; parts <= 4 && wbytes > (3 << 20); parts *= 2)
            if (p.tiles_n % parts == 0 && wbytes / parts <= (5 << 19)) { p.ngroup = p.tiles_n / parts; break; }
    }
    if (modcr_knob_set("MODCR_GEMM_NGROUP")) p.ngroup = modcr_knob_int("MODCR_GEMM_NGROUP", 0);       // tuning build only
    if (p.ngroup < 0 || (p.ngroup > 0 && (p.tiles_n % p.ngroup) != 0)) p.ngroup = 0;                  // (a group count that does not divide: plain walk)
    // persistent: one workgroup per CU (a multiple of 8 so a workgroup's tiles stay on one XCD's chunk)
    const int nwg = p.tiles_m * p.tiles_n * (p.k_tiles_per_split ? (p.K >> 6) / p.k_tiles_per_split : 1);
    const int ncu = modcr_device_cus();
    const int grid = nwg <= ncu ? nwg : (ncu & ~7);
    hipLaunchKernelGGL((linear_bf16_p8_kernel<ACT, RES, OUT, DIRECT, TN, SPEC>), dim3(grid), dim3(512), P8::SMEM, st, p);
    return modcr_check_launch("linear_bf16_p8");
}
#ifdef MODCR_TUNING
// the dual-workgroup kernel (linear_bf16_d4_kernel): whole 256 x 128 tiles, K a multiple of 192
inline bool d4_shape(const LinearArgs& p) {
    return (p.M % 256) == 0 && (p.N % 128) == 0 && (p.K % 192) == 0 && p.K >= 192 && !p.k_tiles_per_split && !p.res && (p.ldc % 8) == 0 &&
           (p.lda % 8) == 0 && (p.ldw % 8) == 0 && (!p.bias || modcr_aligned16(p.bias)) &&
           (int64_t)p.lda * 64 * 2 < (1ll << 31) && (int64_t)p.ldw * 64 * 2 < (1ll << 31) && (int64_t)p.ldc * 32 < (1ll << 31);
}
template <int ACT, int OUT>
int launch_d4(LinearArgs p, hipStream_t st) {
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bf16_d4_kernel<ACT, OUT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, D4::SMEM);
        if (e != hipSuccess) {
            modcr_set_error("linear (d4): cannot reserve %d bytes of LDS: %s", D4::SMEM, hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    p.tiles_m = p.M / 256;
    p.tiles_n = p.N / 128;
    // column groups as in launch_p8d (a group's weight slice <= 2.5 MB stays in the XCD's L2)
    p.ngroup = 0;
    if (p.tiles_m >= 64) {
        const int64_t wbytes = (int64_t)p.N * p.K * 2;
        for (int parts = 2; parts <= 4 && wbytes > (3 << 20); parts *= 2)
            if (p.tiles_n % parts == 0 && wbytes / parts <= (5 << 19)) { p.ngroup = p.tiles_n / parts; break; }
    }
    if (modcr_knob_set("MODCR_GEMM_NGROUP")) p.ngroup = modcr_knob_int("MODCR_GEMM_NGROUP", 0);       // tuning build only
    if (p.ngroup < 0 || (p.ngroup > 0 && (p.tiles_n % p.ngroup) != 0)) p.ngroup = 0;
    p.order = modcr_knob_int("MODCR_GEMM_ORDER", 0);                      // tuning build only
    const int nwg = p.tiles_m * p.tiles_n;
    const int slots = 2 * (modcr_device_cus() & ~7);          // two workgroups per CU
    const int grid = nwg <= slots ? nwg : slots;
    hipLaunchKernelGGL((linear_bf16_d4_kernel<ACT, OUT>), dim3(grid), dim3(D4::NT), D4::SMEM, st, p);
    return modcr_check_launch("linear_bf16_d4");
}
#endif  // MODCR_TUNING (d4)
template <int ACT, int RES, int OUT>
int launch_p8(const LinearArgs& p, hipStream_t st) {
    const int direct = modcr_knob_int("MODCR_GEMM_DIRECT", 1);            // tuning build only
    // the register epilogue reads / writes 8- and 16-byte pieces at column offsets that are multiples of 4
    const bool ok = (p.ldc % 8) == 0 && (!p.res || (p.ldr % 4) == 0) && (!p.bias || modcr_aligned16(p.bias));
    // seamless ring + stores behind vmcnt(0) (see the kernel): 16-bit output without a residual
    if constexpr (RES == 0 && OUT != MODCR_F32 && (ACT == MODCR_ACT_NONE || ACT == MODCR_ACT_GELU || ACT == MODCR_ACT_TANH)) {
#ifdef MODCR_TUNING
        if (direct && ok && d4_shape(p) && modcr_knob_int("MODCR_GEMM_D4", 0)) return launch_d4<ACT, OUT>(p, st);      // A/B only
#endif
        if (direct && ok && !p.k_tiles_per_split && (p.M % 256) == 0 && modcr_knob_int("MODCR_GEMM_SPEC", 1))
            return launch_p8d<ACT, RES, OUT, 1, 0, 1>(p, st);
    }
    return (((direct && RES == 0) || direct == 2) && ok) ? launch_p8d<ACT, RES, OUT, 1>(p, st) : launch_p8d<ACT, RES, OUT, 0>(p, st);   // see launch_t192
}
// shapes the half-tile kernel takes
bool p8_ok(const LinearArgs& p) {
    if (modcr_knob_set("MODCR_GEMM_NO_P8")) return false;                // tuning build only
    if (p.k_tiles_per_split) {       // split-K: equal even splits, plain fp32 partials
        const int kps = p.k_tiles_per_split, nkt = p.K >> 6;
        if ((kps & 1) || kps < 4 || (p.K & 63) || nkt % kps || p.bias || p.res || p.act != MODCR_ACT_NONE || p.out_dtype != MODCR_F32) return false;
    }
    if (p.M < 256 || (p.N % 256) != 0 || (p.K % 128) != 0 || p.K < 256) return false;
    if ((int64_t)p.M * p.lda >= (1ll << 31) || (int64_t)256 * p.ldw >= (1ll << 31)) return false;   // 32-bit byte offsets
    if ((p.ldc % 8) != 0 || !modcr_aligned16(p.C)) return false;
    if (p.res && ((p.ldr % 8) != 0 || !modcr_aligned16(p.res) || (p.M % 8) != 0)) return false;
    return true;
}

template <typename T, int ACT, int RES, int OUT>
int launch_linear(LinearArgs p, hipStream_t st) {
    static bool configured_dev[MODCR_MAX_DEV] = {};
    bool& configured = configured_dev[modcr_device_index()];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bf16_kernel<T, ACT, RES, OUT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, T::SMEM);
        if (e != hipSuccess) {
            modcr_set_error("linear: cannot reserve %d bytes of LDS: %s", T::SMEM, hipGetErrorString(e));
            return MODCR_ERR_LAUNCH;
        }
        configured = true;
    }
    p.tiles_m = (p.M + T::BM - 1) / T::BM;
    p.tiles_n = (p.N + T::BN - 1) / T::BN;
    p.order = modcr_knob_int("MODCR_GEMM_ORDER", 0);                      // tuning build only
    const int splits = p.k_tiles_per_split ? (p.K / BK + p.k_tiles_per_split - 1) / p.k_tiles_per_split : 1;
    hipLaunchKernelGGL((linear_bf16_kernel<T, ACT, RES, OUT>), dim3(p.tiles_m * p.tiles_n, splits), dim3(T::NT),
                       T::SMEM, st, p);
    return modcr_check_launch("linear_bf16");
}

// Tile choice = arithmetic intensity x how well the grid fills whole rounds of the 256 CUs.
// 256x256 and 192x256 run one workgroup per CU, 128x128 two.
inline bool p8_shape(const LinearArgs& p) {
    return !p.k_tiles_per_split && p.M >= 256 && (p.N % 256) == 0 && (p.K % 128) == 0 && p.K >= 256;
}
int choose_tile(const LinearArgs& p) {
    const int force = modcr_knob_int("MODCR_GEMM_TILE", 0);               // tuning build only
    if (force == 128 || p.M < 192 || p.N < 256) return 128;
    if (force == 256 || force == 192) return force;
    const int splits = p.k_tiles_per_split ? (p.K / BK + p.k_tiles_per_split - 1) / p.k_tiles_per_split : 1;
    auto eff = [&](int bm, int bn, int slots) {
        const int64_t t = (int64_t)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * splits;
        const int64_t rounds = (t + slots - 1) / slots;
        const double useful = (double)p.M * p.N / ((double)((p.M + bm - 1) / bm) * bm * (double)((p.N + bn - 1) / bn) * bn);
        return useful * (double)t / (double)(rounds * slots);
    };
    // the persistent 256 x 256 kernel runs ~1.0 PF where the older 192 x 256 / 128 x 128 ring kernels reach 0.6-0.7:
    // they must win on whole rounds by that margin
    const bool fast_l = p8_shape(p);
    const double eL = eff(256, 256, 256), eM = eff(192, 256, 256) * (fast_l ? 0.70 : 0.97), eS = eff(128, 128, 512) * (fast_l ? 0.60 : 0.85);
    if (eL >= eM && eL >= eS) return 256;
    return eM >= eS ? 192 : 128;
}

template <int ACT, int RES, int OUT>
int dispatch_tile(const LinearArgs& p, hipStream_t st) {
    const int knob = modcr_knob_int("MODCR_GEMM_T192", -1);               // tuning build only: 0 off, 1 force
    if (knob != 0 && t192_ok(p)) {
        // whole-round efficiency of the persistent grids (one workgroup per CU).  At equal efficiency the
        // 256 x 256 kernel is ~5 % faster (measured at M = 46080, N = 3072, K = 768): this one wins on rounds only.
        auto eff = [&](int bm, int bn) {
            const int64_t t = (int64_t)((p.M + bm - 1) / bm) * (p.N / bn);
            return (double)p.M * p.N / ((double)((t + 255) / 256) * 256 * bm * bn);
        };
        const bool p8_fits = (p.N % 256) == 0 && p.M >= 256;
        const int64_t t192 = (int64_t)((p.M + 191) / 192) * (p.N / 384);       // small problems: the 128 x 128 tiles fill more CUs
        if (knob == 1 || (t192 >= 192 && (!p8_fits || eff(192, 384) >= eff(256, 256) * 1.04))) return launch_t192<ACT, RES, OUT>(p, st);
    }
    switch (choose_tile(p)) {
        case 256: if (p8_ok(p)) return launch_p8<ACT, RES, OUT>(p, st);
                  return launch_linear<TileL, ACT, RES, OUT>(p, st);
        case 192: return launch_linear<TileM, ACT, RES, OUT>(p, st);
    }
    return launch_linear<TileS, ACT, RES, OUT>(p, st);
}
template <int ACT, int RES>
int dispatch_out(const LinearArgs& p, hipStream_t st) {
    if (p.out_dtype == MODCR_F16) {      // fp16 rows feed a LayerNorm pass: plain products only (no activation, no / bf16 residual)
        if constexpr (ACT == MODCR_ACT_NONE && RES != 2) return dispatch_tile<ACT, RES, MODCR_F16>(p, st);
        modcr_set_error("linear_fwd: fp16 output is implemented for act = none without an fp32 residual");
        return MODCR_ERR_UNSUPPORTED;
    }
    return p.out_dtype == MODCR_BF16 ? dispatch_tile<ACT, RES, MODCR_BF16>(p, st)
                                     : dispatch_tile<ACT, RES, MODCR_F32>(p, st);
}
template <int ACT>
int dispatch_res(const LinearArgs& p, hipStream_t st) {
    if (!p.res) return dispatch_out<ACT, 0>(p, st);
    return p.res_dtype == MODCR_BF16 ? dispatch_out<ACT, 1>(p, st) : dispatch_out<ACT, 2>(p, st);
}
int dispatch_linear(const LinearArgs& p, hipStream_t st) {
    switch (p.act) {
        case MODCR_ACT_NONE: return dispatch_res<MODCR_ACT_NONE>(p, st);
        case MODCR_ACT_GELU: return dispatch_res<MODCR_ACT_GELU>(p, st);
        case MODCR_ACT_TANH: return dispatch_res<MODCR_ACT_TANH>(p, st);
    }
    modcr_set_error("linear_fwd: unknown activation %d", p.act);
    return MODCR_ERR_INVALID;
}

// ---- generic strided fp32 product:  C[m,n] = act(sum_k A(m,k) B(k,n) + bias[n]) (+res), (+C if accumulate)
struct GemmF32Args {
    const void* A; int64_t sam, sak; int a_dtype;
    const void* B; int64_t sbk, sbn; int b_dtype;
    const float* bias;
    const void* res; int64_t ldr; int res_dtype;
    void* C; int64_t ldc; int out_dtype;
    int M, N, K, act, accumulate;
};

__device__ __forceinline__ float ld_any(const void* p, int64_t off, int dt) {
    return dt == MODCR_BF16 ? (float)reinterpret_cast<const bf16*>(p)[off]
                            : reinterpret_cast<const float*>(p)[off];
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args p) {
    __shared__ float sA[16][65];
    __shared__ float sB[16][65];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = tid & 15, ty = tid >> 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    // staging: 64x16 elements per operand, 4 per thread
    const int lk = tid & 15, lr = tid >> 4;   // k fastest (good when k is the contiguous dim)
    const int lk2 = tid >> 6, lr2 = tid & 63; // row fastest (good when m / n is contiguous)
    const bool a_kfast = p.sak == 1, b_kfast = p.sbk == 1;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int rr, kk;
            if (a_kfast) { rr = lr + 16 * i; kk = lk; } else { rr = lr2; kk = lk2 + 4 * i; }
            const int m = m0 + rr, k = k0 + kk;
            sA[kk][rr] = (m < p.M && k < p.K) ? ld_any(p.A, m * p.sam + k * p.sak, p.a_dtype) : 0.f;
            if (b_kfast) { rr = lr + 16 * i; kk = lk; } else { rr = lr2; kk = lk2 + 4 * i; }
            const int n = n0 + rr;
            const int kb = k0 + kk;
            sB[kk][rr] = (n < p.N && kb < p.K) ? ld_any(p.B, kb * p.sbk + n * p.sbn, p.b_dtype) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = sA[kk][ty + 16 * i]; b[i] = sB[kk][tx + 16 * i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 16 * i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx + 16 * j;
            if (n >= p.N) continue;
            float v = acc[i][j] + (p.bias ? p.bias[n] : 0.f);
            v = act_apply_exact(v, p.act);
            if (p.res) v += ld_any(p.res, (int64_t)m * p.ldr + n, p.res_dtype);
            const int64_t o = (int64_t)m * p.ldc + n;
            if (p.out_dtype == MODCR_BF16) {
                reinterpret_cast<bf16*>(p.C)[o] = (bf16)v;
            } else {
                float* c = reinterpret_cast<float*>(p.C);
                c[o] = p.accumulate ? c[o] + v : v;
            }
        }
    }
}

// column sums of dY [M,N] -> db[N] (fp32), one block per 64 columns
__global__ __launch_bounds__(256) void colsum_kernel(const void* dY, int64_t ld, int dt, float* db,
                                                     int M, int N, int accumulate) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int w = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N)
        for (int m = w; m < M; m += 4) s += ld_any(dY, (int64_t)m * ld + c, dt);
    part[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < N) {
        s = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        db[c] = accumulate ? db[c] + s : s;
    }
}

// column sums of a bf16 [M,N] matrix (N % 8 == 0, 16-byte rows): block = 128 rows x up to 2048 columns, a thread sums one
// 16-byte chunk column over the rows, one atomic per column per block (db zeroed by the caller).  (A version with 64-row
// blocks and four row streams per thread was slower: twice the atomics on the same N addresses.)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16* dY, int64_t ld, float* db, int M, int N) {
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc * 8 >= N) return;
    const int r0 = blockIdx.y * 128, r1 = min(r0 + 128, M);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16* src = dY + cc * 8;
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (int64_t)r * ld);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(db + cc * 8 + e, s[e]);
}

// the same over 256-row x 256-column blocks: 32 chunk columns x 8 row lanes per workgroup, eight 16-byte loads in flight per thread,
// one LDS reduction over the row lanes, one PLAIN store per column and block into part[blockIdx.y][N] (the bias gradient where dY is not
// transposed: the swapped half-TN form of modcr_linear_bwd_weight; the partials are folded, in block order, by extra blocks of the
// split-K reduction that runs anyway -- round 5: was a memset + one float atomic per column and block)
__global__ __launch_bounds__(256) void colsum_bf16_block_kernel(const bf16* dY, int64_t ld, float* part_out, int M, int N) {
    __shared__ float part[8][256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 256 + tx * 8;
    const int r0 = blockIdx.y * 256;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < N) {
        const bf16* src = dY + c;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int r = r0 + ty + 8 * i;
            if (r < M) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + (int64_t)r * ld);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[ty][tx * 8 + e] = s[e];
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
        part_out[(int64_t)blockIdx.y * N + col] = t;
    }
}

// few output columns (the scorers' Linear(., 1): [512 x 1536] . [1 x 1536]^T filled 8 workgroups of the tile kernel for 96
// barrier-separated K steps, 127 us): one wave per output row, lanes strided over K, K contiguous in both operands
__global__ __launch_bounds__(256) void rowdot_f32_kernel(GemmF32Args p) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= p.M) return;
    for (int n = 0; n < p.N; ++n) {
        float s = 0.f;
        for (int k = lane; k < p.K; k += 64)
            s = fmaf(ld_any(p.A, (int64_t)m * p.sam + k, p.a_dtype), ld_any(p.B, (int64_t)n * p.sbn + k, p.b_dtype), s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            float v = s + (p.bias ? p.bias[n] : 0.f);
            v = act_apply_exact(v, p.act);
            if (p.res) v += ld_any(p.res, (int64_t)m * p.ldr + n, p.res_dtype);
            const int64_t o = (int64_t)m * p.ldc + n;
            if (p.out_dtype == MODCR_BF16) {
                reinterpret_cast<bf16*>(p.C)[o] = (bf16)v;
            } else {
                float* c = reinterpret_cast<float*>(p.C);
                c[o] = p.accumulate ? c[o] + v : v;
            }
        }
    }
}

int launch_gemm_f32(const GemmF32Args& a, hipStream_t st) {
    if (a.N <= 4 && a.sak == 1 && a.sbk == 1 && a.M >= 64) {
        hipLaunchKernelGGL(rowdot_f32_kernel, dim3((a.M + 3) / 4), dim3(256), 0, st, a);
        return modcr_check_launch("rowdot_f32");
    }
    dim3 grid((a.N + 63) / 64, (a.M + 63) / 64);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, a);
    return modcr_check_launch("gemm_f32");
}

}  // namespace

extern "C" int modcr_linear_fwd(const void* A, int64_t lda, const void* W, int64_t ldw,
                                const float* bias, const void* residual, int64_t ldr,
                                int32_t res_dtype, void* C, int64_t ldc, int32_t M, int32_t N,
                                int32_t K, int32_t act, int32_t dtype, int32_t out_dtype,
                                modcr_stream_t stream) {
    MODCR_REQUIRE(A && W && C, "linear_fwd: null pointer");
    MODCR_REQUIRE(M > 0 && N > 0 && K > 0, "linear_fwd: bad shape M=%d N=%d K=%d", M, N, K);
    MODCR_REQUIRE(lda >= K && ldw >= K && ldc >= N, "linear_fwd: strides smaller than the row");
    MODCR_REQUIRE(!residual || ldr >= N, "linear_fwd: residual stride");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MODCR_BF16) {
        MODCR_REQUIRE((K % 64) == 0 && (lda % 8) == 0 && (ldw % 8) == 0,
                      "linear_fwd(bf16): K=%d must be a multiple of 64, lda=%lld ldw=%lld multiples of 8 "
                      "(zero-pad with modcr_cast_pad)", K, (long long)lda, (long long)ldw);
        MODCR_REQUIRE(modcr_aligned16(A) && modcr_aligned16(W), "linear_fwd(bf16): 16-byte alignment");
        LinearArgs p;
        p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.bias = bias;
        p.res = residual; p.ldr = ldr; p.res_dtype = res_dtype; p.C = C; p.ldc = ldc;
        p.out_dtype = out_dtype; p.M = M; p.N = N; p.K = K; p.act = act;
        p.tiles_m = p.tiles_n = 0;
        p.vec_ok = (ldc % 4 == 0) && modcr_aligned16(C) && (!residual || ((ldr % 4 == 0) && modcr_aligned16(residual)));
        p.k_tiles_per_split = 0; p.split_stride = 0;
        return dispatch_linear(p, st);
    }
    MODCR_REQUIRE(dtype == MODCR_F32, "linear_fwd: unknown dtype %d", dtype);
    GemmF32Args a;
    a.A = A; a.sam = lda; a.sak = 1; a.a_dtype = MODCR_F32;
    a.B = W; a.sbk = 1; a.sbn = ldw; a.b_dtype = MODCR_F32;
    a.bias = bias; a.res = residual; a.ldr = ldr; a.res_dtype = res_dtype;
    a.C = C; a.ldc = ldc; a.out_dtype = out_dtype; a.M = M; a.N = N; a.K = K; a.act = act;
    a.accumulate = 0;
    return launch_gemm_f32(a, st);
}

extern "C" int modcr_ffn_up_gelu_fwd(const void* x, const void* w1, const float* b1, void* out,
                                     int32_t M, int32_t H, int32_t I, int32_t dtype,
                                     modcr_stream_t stream) {
    return modcr_linear_fwd(x, H, w1, H, b1, nullptr, 0, 0, out, I, M, I, H, MODCR_ACT_GELU, dtype,
                            dtype, stream);
}

namespace {

// dst[n][m] = src[m][n] (bf16 out), 64x64 tiles through LDS; columns m in [M, Mp) are zero-filled
template <typename TS>
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const TS* src, int64_t lds_, bf16* dst, int64_t ldd,
                                                                int M, int N, int Mp) {
    __shared__ float tile[64][65];
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int m = m0 + i, n = n0 + tx;
        tile[i][tx] = (m < M && n < N) ? to_f32(src[(int64_t)m * lds_ + n]) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int n = n0 + i, m = m0 + tx;
        if (n < N && m < Mp) dst[(int64_t)n * ldd + m] = (bf16)tile[tx][i];
    }
}

// The bias gradient of a weight-gradient product, deterministic and without a memset: the pass that walks dY anyway (its transpose,
// or the column-sum pass of the swapped form) leaves one row of N partial sums per row block in the workspace, and EXTRA blocks of
// the split-K reduction that runs behind the product fold them in block order: db[n] (+)= sum_b dbp[b][n].  (Rounds 1-4: a
// hipMemsetAsync of db + one float atomic per column and block -- 122 fill launches per config-3 step, sums that differed in the last
// bit from run to run.)
struct DbFold { const float* part; int nblk, N; float* db; int accumulate; };
constexpr int DBF_COLS = 16;                // columns per fold block: 16 columns x 16 row lanes (one thread per column measured +17 us per
                                            // call at 360 row blocks: a serial chain of loads; this way a lane sums nblk / 16 rows)
__device__ __forceinline__ void db_fold_block(const DbFold& f, int blk) {
    __shared__ float red[16][DBF_COLS + 1];
    const int tx = threadIdx.x & (DBF_COLS - 1), ty = threadIdx.x >> 4;
    const int n = blk * DBF_COLS + tx;
    float s = 0.f;
    if (n < f.N) {
#pragma unroll 8
        for (int b = ty; b < f.nblk; b += 16) s += f.part[(int64_t)b * f.N + n];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < f.N) {
        float t = f.accumulate ? f.db[n] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][tx];           // fixed order: the sum is reproducible
        f.db[n] = t;
    }
}
__global__ __launch_bounds__(256) void db_fold_kernel(DbFold f) { db_fold_block(f, blockIdx.x); }

// out[i] (+)= sum_s partial[s][i]; blocks nb_main .. : the bias-gradient fold above (f.db == NULL: none)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* part, int splits, int64_t stride, float* out, int64_t n,
                                                              int accumulate, int nb_main, DbFold f) {
    if ((int)blockIdx.x >= nb_main) { db_fold_block(f, (int)blockIdx.x - nb_main); return; }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = accumulate ? out[i] : 0.f;
    for (int k = 0; k < splits; ++k) s += part[(int64_t)k * stride + i];
    out[i] = s;
}

// out[n][k] (+)= sum_s partial[s][k][n]: the split-K reduction of a product that was formed transposed (dW^T = X^T dY), 32 x 32
// tiles through LDS; R = rows of the partials (k), C = their columns (n)
__global__ __launch_bounds__(256) void reduce_partials_transposed_kernel(const float* part, int splits, int64_t stride, float* out, int R, int C,
                                                                         int accumulate) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        float s = 0.f;
        if (r < R && c < C)
            for (int k = 0; k < splits; ++k) s += part[(int64_t)k * stride + (int64_t)r * C + c];
        tile[ty + 8 * i][tx] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < R && c < C) {
            float* o = out + (int64_t)c * R + r;
            *o = (accumulate ? *o : 0.f) + tile[tx][ty + 8 * i];
        }
    }
}

// out[m][n] = act(sum_s partial[s][m][n] + bias[n]) (fp32 or bf16 out, row stride ldc): epilogue of the split-K forward
template <typename TO>
__global__ __launch_bounds__(256) void reduce_bias_act_kernel(const float* part, int splits, int64_t stride, const float* bias,
                                                              TO* out, int64_t ldc, int M, int N, int act) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= (int64_t)M * N) return;
    const int m = (int)(i4 / N), n = (int)(i4 % N);                 // N % 4 == 0: the four values share a row
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < splits; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(part + (int64_t)k * stride + i4);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    float v[4] = {s[0], s[1], s[2], s[3]};
    const float b4[4] = {bias ? bias[n] : 0.f, bias ? bias[n + 1] : 0.f, bias ? bias[n + 2] : 0.f, bias ? bias[n + 3] : 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = act_apply_exact(v[e] + b4[e], act);
    TO* o = out + (int64_t)m * ldc + n;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = from_f32<TO>(v[e]);
}

// out[r] (+)= sum_c x[r][c], one wave per row (x bf16, row stride ld)
__global__ __launch_bounds__(256) void rowsum_bf16_kernel(const bf16* x, int64_t ld, float* out, int rows, int cols,
                                                          int accumulate) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bf16* row = x + (int64_t)r * ld;
    float s = 0.f;
    for (int c = lane * 8; c + 7 < cols; c += 512) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(row + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)v[j];
    }
    for (int c = (cols & ~7) + lane; c < cols; c += 64) s += (float)row[c];
    s = wave_sum(s);
    if (lane == 0) out[r] = accumulate ? out[r] + s : s;
}

// dst[n][m] = src[m][n] (bf16 out) in 64 x 64 tiles: 16-byte global loads into a row-major bf16 LDS tile, read back
// column-major with ds_read_b64_tr_b16 (a lane receives 2 x 4 consecutive m of one column n = one 16-byte store; the
// four lane groups of a wave and the two passes complete a row's 128-byte line).  Rows m in [M, Mp) are zero-filled;
// rowsum (optional): [gridDim.x][N] partial sums over this block's rows m of src[m][n], plain stores (the bias gradient of the dW
// product that consumes dst; folded by DbFold blocks of the split-K reduction).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <typename TS>
__global__ __launch_bounds__(256) void transpose64_kernel(const TS* src, int64_t lds_, bf16* dst, int64_t ldd, int M, int N,
                                                          float* rowsum) {
    constexpr int RS = 144;                                 // LDS row stride in bytes (128 + 16: rows 4 apart on other banks)
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * RS];
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int ch = tid + 256 * it, r = ch >> 3, c = ch & 7;
        const int m = m0 + r, n = n0 + 8 * c;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)0.f;
        if (m < M && n < N) {
            const TS* sp = src + (int64_t)m * lds_ + n;
            if constexpr (sizeof(TS) == 2) {
                v = *reinterpret_cast<const bf16x8*>(sp);
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (bf16)a[e]; v[4 + e] = (bf16)b[e]; }
            }
        }
        *reinterpret_cast<bf16x8*>(tile + r * RS + 16 * c) = v;
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)tile;
    const int n = n0 + 16 * w + i;
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int C = g + 4 * it;                           // chunk of 8 rows m
        const unsigned a0 = lds0 + (8 * C + q) * RS + (16 * w + 4 * pp) * 2;
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:576\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi) : "v"(a0) : "memory");
        if (n < N) *reinterpret_cast<uint4*>(dst + (int64_t)n * ldd + m0 + 8 * C) = make_uint4(lo[0], lo[1], hi[0], hi[1]);
        if (rowsum) {
            const unsigned wv[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __uint_as_float(wv[e] << 16) + __uint_as_float(wv[e] & 0xffff0000u);
        }
    }
    if (rowsum) {
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0 && n < N) rowsum[(int64_t)blockIdx.x * N + n] = s;
    }
}

// The same on 256 x 64 source blocks (m x n): eight 16-byte loads per thread in flight before the LDS pass and 512-byte runs
// per destination row instead of 128 -- the [92160 x 768] operand transposes of the weight-gradient products ran at 2.7 TB/s
// (103 us) with one 64 x 64 tile per workgroup.
template <typename TS>
__global__ __launch_bounds__(256) void transpose256_kernel(const TS* src, int64_t lds_, bf16* dst, int64_t ldd, int M, int N,
                                                           float* rowsum) {
    constexpr int RS = 144;
    __shared__ __attribute__((aligned(16))) unsigned char tile[256 * RS];
    const int m0 = blockIdx.x * 256, n0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    bf16x8 v[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int ch = tid + 256 * it, r = ch >> 3, c = ch & 7;
        const int m = m0 + r, n = n0 + 8 * c;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[it][e] = (bf16)0.f;
        if (m < M && n < N) {
            const TS* sp = src + (int64_t)m * lds_ + n;
            if constexpr (sizeof(TS) == 2) {
                v[it] = *reinterpret_cast<const bf16x8*>(sp);
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[it][e] = (bf16)a[e]; v[it][4 + e] = (bf16)b[e]; }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int ch = tid + 256 * it, r = ch >> 3, c = ch & 7;
        *reinterpret_cast<bf16x8*>(tile + r * RS + 16 * c) = v[it];
    }
    __syncthreads();
    const int lane = tid & 63, w = tid >> 6, g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)tile;
    const int n = n0 + 16 * w + i;
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int C = g + 4 * it;                           // chunk of 8 rows m
        const unsigned a0 = lds0 + (8 * C + q) * RS + (16 * w + 4 * pp) * 2;
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:576\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi) : "v"(a0) : "memory");
        if (n < N) *reinterpret_cast<uint4*>(dst + (int64_t)n * ldd + m0 + 8 * C) = make_uint4(lo[0], lo[1], hi[0], hi[1]);
        if (rowsum) {
            const unsigned wv[4] = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __uint_as_float(wv[e] << 16) + __uint_as_float(wv[e] & 0xffff0000u);
        }
    }
    if (rowsum) {
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (g == 0 && n < N) rowsum[(int64_t)blockIdx.x * N + n] = s;
    }
}

// rowsum: NULL, or fp32 [*nblk][N] that receives the per-row-block partial column sums of src (*nblk = the row blocks used: <= Mp / 64)
int transpose_to_bf16(const void* src, int src_dtype, int64_t lds_, bf16* dst, int64_t ldd, int M, int N, int Mp,
                      hipStream_t st, float* rowsum = nullptr, int* nblk = nullptr) {
    dim3 grid((Mp + 63) / 64, (N + 63) / 64);
    const bool vec = (N % 8) == 0 && (lds_ % 8) == 0 && (Mp % 64) == 0 && (ldd % 8) == 0 && modcr_aligned16(src) && modcr_aligned16(dst);
    if (nblk) *nblk = (vec && (Mp % 256) == 0 && Mp >= 4096) ? Mp / 256 : (Mp + 63) / 64;
    if (vec && (Mp % 256) == 0 && Mp >= 4096) {
        const dim3 grid4(Mp / 256, (N + 63) / 64);
        if (src_dtype == MODCR_BF16)
            hipLaunchKernelGGL((transpose256_kernel<bf16>), grid4, dim3(256), 0, st, (const bf16*)src, lds_, dst, ldd, M, N, rowsum);
        else
            hipLaunchKernelGGL((transpose256_kernel<float>), grid4, dim3(256), 0, st, (const float*)src, lds_, dst, ldd, M, N, rowsum);
        return modcr_check_launch("transpose256");
    }
    if (vec) {
        if (src_dtype == MODCR_BF16)
            hipLaunchKernelGGL((transpose64_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)src, lds_, dst, ldd, M, N, rowsum);
        else
            hipLaunchKernelGGL((transpose64_kernel<float>), grid, dim3(256), 0, st, (const float*)src, lds_, dst, ldd, M, N, rowsum);
        return modcr_check_launch("transpose64");
    }
    MODCR_REQUIRE(!rowsum, "transpose_to_bf16: fused row sums need the vector path");
    if (src_dtype == MODCR_BF16)
        hipLaunchKernelGGL((transpose_to_bf16_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)src, lds_, dst, ldd, M, N, Mp);
    else
        hipLaunchKernelGGL((transpose_to_bf16_kernel<float>), grid, dim3(256), 0, st, (const float*)src, lds_, dst, ldd, M, N, Mp);
    return modcr_check_launch("transpose_to_bf16");
}

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct BwdWeightPlan { int64_t Mp; int splits, kps; int64_t off_xt, off_part, off_dbp, total; };
BwdWeightPlan plan_bwd_weight(int M, int N, int K) {
    BwdWeightPlan p;
    const int tiles = (int)(((N + 255) / 256) * (int64_t)((K + 255) / 256));
    const int ktiles = (int)(align_up(M, 64) / 64);
    // Equal splits of an even number (>= 4) of 64-token K-tiles (what the persistent 256 x 256 kernel takes); the token dim is
    // zero-padded up to splits * kps * 64 by the transposes.  The split count is the one that minimises
    //   rounds of 256 workgroups x (K-tiles per work item + ~6 K-tiles' worth of prologue / fp32 partial store) + the reduce pass:
    // "about two rounds" (round 1) gave 9 tiles x 32 splits = 288 items = 2 rounds at 56 % for the [768 x 768] products and
    // 27 x 19 = 513 = 3 rounds for [2304 x 768]; 28 and 9 splits fill one round each with 1.1 / 2.1 times longer items.
    int best = 1, best_kps = 0;
    double best_cost = 1e30;
    for (int sp = 1; sp <= 32 && sp <= ktiles; ++sp) {
        int kps = (ktiles + sp - 1) / sp;
        if (kps & 1) ++kps;
        if (kps < 4) kps = 4;
        if (ktiles >= 64) kps = (kps + 3) & ~3;                 // padded token count % 256 == 0: the 256 x 64 operand transposes
        const int s2 = (ktiles + kps - 1) / kps;
        if (s2 != sp && sp > 1) continue;                       // the same plan as a smaller candidate
        const int64_t items = (int64_t)tiles * s2;
        const double rounds = (double)((items + 255) / 256);
        const double cost = rounds * (kps + 6.0) + 0.04 * (double)items;
        if (cost < best_cost) { best_cost = cost; best = s2; best_kps = kps; }
    }
    p.kps = best_kps;
    p.splits = best;
    p.Mp = (int64_t)p.splits * p.kps * 64;
    const int64_t dyt = align_up((int64_t)N * p.Mp * 2, 256);
    const int64_t xt = align_up((int64_t)K * p.Mp * 2, 256);
    p.off_xt = dyt;
    p.off_part = dyt + xt;
    p.off_dbp = p.off_part + align_up((int64_t)p.splits * N * K * 4, 256);
    // bias-gradient partials: one row per 64-token block (upper bound of both passes that write them) x the wider of the two operands
    p.total = p.off_dbp + align_up((p.Mp / 64) * (int64_t)(N > K ? N : K) * 4, 256);
    return p;
}

}  // namespace

// ---- split-K forward for GEMMs with few rows (the trainable heads at M = 256: 1 x N/256 output tiles fill 3-20 of the
// 256 CUs): the persistent 256 x 256 kernel over (split, tile) work items into fp32 partials, then one pass that sums
// them and applies bias / activation.  The plan is a pure function of the shape; 0 bytes = not applicable / not worth it.
namespace {
struct SplitKPlan { int splits, kps; int64_t bytes; };
SplitKPlan plan_splitk_fwd(int M, int N, int K) {
    SplitKPlan pl = {0, 0, 0};
    if (M < 256 || M > 1024 || (N % 256) != 0 || (K % 128) != 0 || K < 1024) return pl;      // 1-4 rows of tiles, long K
    const int tiles = ((M + 255) / 256) * (N / 256), nkt = K / 64;
    if (tiles >= 96) return pl;
    // split count by whole rounds of 256 workgroups (as plan_bwd_weight): rounds x (K-tiles per item + ~6 of fixed cost)
    int best = 0;
    double best_cost = 1e30;
    for (int sp = 2; sp <= 64 && sp <= nkt / 4; ++sp) {
        if (nkt % sp) continue;
        const int kps = nkt / sp;
        if ((kps & 1) || kps < 4) continue;
        const double cost = (double)((sp * tiles + 255) / 256) * (kps + 6.0) + 0.04 * sp * tiles;
        if (cost < best_cost) { best_cost = cost; best = sp; }
    }
    if (!best) return pl;
    pl.splits = best; pl.kps = nkt / best; pl.bytes = (int64_t)best * M * N * 4;
    return pl;
}
}  // namespace

extern "C" int64_t modcr_linear_splitk_workspace(int32_t M, int32_t N, int32_t K) { return plan_splitk_fwd(M, N, K).bytes; }

extern "C" int modcr_linear_splitk_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                                       int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, int32_t out_dtype,
                                       void* workspace, int64_t workspace_bytes, modcr_stream_t stream) {
    MODCR_REQUIRE(A && W && C && workspace, "linear_splitk_fwd: null pointer");
    const SplitKPlan pl = plan_splitk_fwd(M, N, K);
    MODCR_REQUIRE(pl.bytes > 0 && workspace_bytes >= pl.bytes, "linear_splitk_fwd: shape M=%d N=%d K=%d has no split-K plan or the workspace is too small", M, N, K);
    MODCR_REQUIRE((lda % 8) == 0 && (ldw % 8) == 0 && modcr_aligned16(A) && modcr_aligned16(W) && modcr_aligned16(workspace),
                  "linear_splitk_fwd: 16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    LinearArgs p;
    p.A = (const bf16*)A; p.lda = lda; p.W = (const bf16*)W; p.ldw = ldw; p.bias = nullptr; p.res = nullptr; p.ldr = 0;
    p.res_dtype = 0; p.C = workspace; p.ldc = N; p.out_dtype = MODCR_F32; p.M = M; p.N = N; p.K = K;
    p.act = MODCR_ACT_NONE; p.tiles_m = p.tiles_n = 0; p.vec_ok = 1;
    p.k_tiles_per_split = pl.kps; p.split_stride = (int64_t)M * N;
    MODCR_REQUIRE(p8_ok(p), "linear_splitk_fwd: shape not supported by the persistent kernel");
    int rc = launch_p8<MODCR_ACT_NONE, 0, MODCR_F32>(p, st);
    if (rc != MODCR_OK) return rc;
    const int64_t n4 = ((int64_t)M * N + 3) / 4;
    const dim3 grid((unsigned)((n4 + 255) / 256));
    if (out_dtype == MODCR_BF16)
        hipLaunchKernelGGL((reduce_bias_act_kernel<bf16>), grid, dim3(256), 0, st, (const float*)workspace, pl.splits, (int64_t)M * N, bias, (bf16*)C, ldc, M, N, act);
    else
        hipLaunchKernelGGL((reduce_bias_act_kernel<float>), grid, dim3(256), 0, st, (const float*)workspace, pl.splits, (int64_t)M * N, bias, (float*)C, ldc, M, N, act);
    return modcr_check_launch("reduce_bias_act");
}

// MFMA route for dW: transpose dY and X to [*, M] bf16 (contraction dim contiguous), NT GEMM with
// split-K into fp32 partials, reduce.  db = row sums of dY^T.
extern "C" int64_t modcr_linear_bwd_weight_workspace(int32_t M, int32_t N, int32_t K) {
    const int64_t a = plan_bwd_weight(M, N, K).total, b = plan_bwd_weight(M, K, N).total;       // (the swapped form of a dY wider than X)
    return a > b ? a : b;
}
extern "C" int64_t modcr_linear_splitk_workspace(int32_t M, int32_t N, int32_t K);
extern "C" int modcr_linear_splitk_fwd(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* C,
                                       int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t act, int32_t out_dtype,
                                       void* workspace, int64_t workspace_bytes, modcr_stream_t stream);
extern "C" int64_t modcr_linear_bwd_input_workspace(int32_t M, int32_t N, int32_t K) {
    const int64_t Np = align_up(N, 64);
    // W^T | bf16 dY | split-K partials of the few-row case (dX [M,K] = dY [M,Np] . W^T: the product (M, K, Np))
    return align_up((int64_t)K * Np * 2, 256) + align_up((int64_t)M * Np * 2, 256) + modcr_linear_splitk_workspace(M, K, (int32_t)Np);
}

extern "C" int modcr_linear_bwd_input(const void* dY, int64_t lddy, int32_t dy_dtype, const void* W,
                                      int64_t ldw, void* dX, int64_t lddx, int32_t M, int32_t N, int32_t K,
                                      int32_t dtype, int32_t out_dtype, void* workspace, int64_t workspace_bytes,
                                      modcr_stream_t stream) {
    return modcr_linear_bwd_input_res(dY, lddy, dy_dtype, W, ldw, nullptr, 0, dX, lddx, M, N, K, dtype, out_dtype, workspace,
                                      workspace_bytes, stream);
}

// dX = dY.W (+ res): `res` fp32 [M,K] (row stride ldr) or NULL is added in the GEMM's epilogue -- the residual-stream gradient
// that the layer backward used to add in a pass of its own (modcr_add).  Library-internal (common.h), not part of the C ABI.
int modcr_linear_bwd_input_res(const void* dY, int64_t lddy, int32_t dy_dtype, const void* W, int64_t ldw, const float* res,
                               int64_t ldr, void* dX, int64_t lddx, int32_t M, int32_t N, int32_t K, int32_t dtype,
                               int32_t out_dtype, void* workspace, int64_t workspace_bytes, modcr_stream_t stream) {
    MODCR_REQUIRE(dY && W && dX && M > 0 && N > 0 && K > 0, "linear_bwd_input: bad arguments");
    const int64_t Np = align_up(N, 64);
    const int64_t off_sk = align_up((int64_t)K * Np * 2, 256) + align_up((int64_t)M * Np * 2, 256);
    const int64_t sk_bytes = modcr_linear_splitk_workspace(M, K, (int32_t)Np);
    const int64_t need = off_sk + sk_bytes;
    if (workspace && workspace_bytes >= need && N >= 64) {
        // dX[M,K] = dY[M,N] . W[N,K] = NT product of dY (bf16, N padded to 64) with W^T [K,N]
        hipStream_t st = (hipStream_t)stream;
        bf16* wt = (bf16*)workspace;
        bf16* dyb = (bf16*)((char*)workspace + align_up((int64_t)K * Np * 2, 256));
        int rc = transpose_to_bf16(W, dtype, ldw, wt, Np, N, K, (int)Np, st);
        if (rc != MODCR_OK) return rc;
        if (dy_dtype == MODCR_F32) {
            rc = modcr_cast_pad((const float*)dY, lddy, dyb, Np, M, N, (int)Np, MODCR_BF16, stream);
        } else {
            // bf16 dY: pad columns by a transposing round trip is overkill; require N % 64 == 0
            MODCR_REQUIRE(N % 64 == 0 && lddy % 8 == 0, "linear_bwd_input: bf16 dY needs N %% 64 == 0");
            dyb = (bf16*)dY;
        }
        if (rc != MODCR_OK) return rc;
        if (!res && sk_bytes && ((dy_dtype == MODCR_F32 ? Np : lddy) % 8) == 0 && modcr_aligned16(dyb))      // few rows: split-K over the chip
            return modcr_linear_splitk_fwd(dyb, dy_dtype == MODCR_F32 ? Np : lddy, wt, Np, nullptr, dX, lddx, M, K, (int)Np,
                                           MODCR_ACT_NONE, out_dtype, (char*)workspace + off_sk, sk_bytes, stream);
        return modcr_linear_fwd(dyb, dy_dtype == MODCR_F32 ? Np : lddy, wt, Np, nullptr, res, ldr, MODCR_F32, dX, lddx, M, K,
                                (int)Np, MODCR_ACT_NONE, MODCR_BF16, out_dtype, stream);
    }
    // dX[m,k] = sum_n dY[m,n] W[n,k]
    GemmF32Args a;
    a.A = dY; a.sam = lddy; a.sak = 1; a.a_dtype = dy_dtype;
    a.B = W; a.sbk = ldw; a.sbn = 1; a.b_dtype = dtype;
    a.bias = nullptr; a.res = res; a.ldr = ldr; a.res_dtype = MODCR_F32;
    a.C = dX; a.ldc = lddx; a.out_dtype = out_dtype; a.M = M; a.N = K; a.K = N; a.act = 0;
    a.accumulate = 0;
    return launch_gemm_f32(a, (hipStream_t)stream);
}

namespace {
// the split-K reduction of a weight-gradient product (+ the bias-gradient fold as extra blocks, f.db != NULL)
int launch_reduce_partials(const float* part, int splits, int64_t nel, float* dW, int accumulate, const DbFold& f, hipStream_t st) {
    const int nb_main = (int)((nel + 255) / 256), nb_fold = f.db ? (f.N + DBF_COLS - 1) / DBF_COLS : 0;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)(nb_main + nb_fold)), dim3(256), 0, st, part, splits, nel, dW, nel, accumulate, nb_main, f);
    return modcr_check_launch("reduce_partials");
}
int launch_db_fold(const DbFold& f, hipStream_t st) {
    hipLaunchKernelGGL(db_fold_kernel, dim3((unsigned)((f.N + DBF_COLS - 1) / DBF_COLS)), dim3(256), 0, st, f);
    return modcr_check_launch("db_fold");
}
}  // namespace

extern "C" int modcr_linear_bwd_weight(const void* dY, int64_t lddy, int32_t dy_dtype, const void* X,
                                       int64_t ldx, float* dW, float* db, int32_t M, int32_t N, int32_t K,
                                       int32_t accumulate, int32_t dtype, void* workspace, int64_t workspace_bytes,
                                       modcr_stream_t stream) {
    MODCR_REQUIRE(dY && X && dW && M > 0 && N > 0 && K > 0, "linear_bwd_weight: bad arguments");
    // TN route (opt-in, MODCR_GEMM_TN=1): both operands stay token-major (no transposes): the persistent 256 x 256 kernel
    // stages [64 tokens][128 features] half-tiles and reads its MFMA fragments with transposed LDS reads.  Needs bf16
    // operands, whole 256 x 256 output tiles and a token count that splits into equal even runs of 64-token K-tiles.
    // Correct (tests) and conflict-free (PMC), but the product itself runs 25 % slower than the row-major form (209 vs
    // 167 us averaged over the encoder shapes, twice the LDS read instructions) and its separate column-sum pass costs
    // what the two transposes it saves cost: 323 vs 320 us per dW end to end -- off by default.
    const int tn_on = modcr_knob_int("MODCR_GEMM_TN", 0);                 // tuning build only
    if (tn_on && workspace && dy_dtype == MODCR_BF16 && dtype == MODCR_BF16 && (N % 256) == 0 && (K % 256) == 0 && (M % 128) == 0 &&
        (lddy % 8) == 0 && (ldx % 8) == 0 && modcr_aligned16(dY) && modcr_aligned16(X) &&
        (int64_t)64 * lddy * 2 + (int64_t)N * 2 < (1ll << 31) && (int64_t)64 * ldx * 2 + (int64_t)K * 2 < (1ll << 31)) {
        const int ktiles = M / 64, tiles = (N / 256) * (K / 256);
        // splits: a divisor of ktiles with an even quotient >= 4, closest to ~2 rounds of 256 work items, at most 32
        int best = 0;
        for (int sp = 1; sp <= 32 && sp <= ktiles; ++sp) {
            if (ktiles % sp) continue;
            const int kps = ktiles / sp;
            if ((kps & 1) || kps < 4) continue;
            if (!best || abs(sp * tiles - 512) < abs(best * tiles - 512)) best = sp;
        }
        if (best && (int64_t)best * N * K * 4 <= workspace_bytes) {
            hipStream_t st = (hipStream_t)stream;
            float* part = (float*)workspace;
            LinearArgs p;
            p.A = (const bf16*)dY; p.lda = lddy; p.W = (const bf16*)X; p.ldw = ldx; p.bias = nullptr; p.res = nullptr; p.ldr = 0;
            p.res_dtype = 0; p.C = part; p.ldc = K; p.out_dtype = MODCR_F32; p.M = N; p.N = K; p.K = M;
            p.act = MODCR_ACT_NONE; p.tiles_m = p.tiles_n = 0; p.vec_ok = 1;
            p.k_tiles_per_split = ktiles / best; p.split_stride = (int64_t)N * K;
            int rc = launch_p8d<MODCR_ACT_NONE, 0, MODCR_F32, 1, 1>(p, st);
            if (rc != MODCR_OK) return rc;
            const int64_t nel = (int64_t)N * K;
            rc = launch_reduce_partials(part, best, nel, dW, accumulate, DbFold{nullptr, 0, 0, nullptr, 0}, st);
            if (rc != MODCR_OK || !db) return rc;
            if (!accumulate && hipMemsetAsync(db, 0, (size_t)N * sizeof(float), st) != hipSuccess) {
                modcr_set_error("linear_bwd_weight: hipMemsetAsync failed");
                return MODCR_ERR_LAUNCH;
            }
            hipLaunchKernelGGL(colsum_bf16_kernel, dim3((N / 8 + 255) / 256, (M + 127) / 128), dim3(256), 0, st, (const bf16*)dY, lddy, db, M, N);
            return modcr_check_launch("colsum_bf16");
        }
    }
    // dY wider than X: the product is formed TRANSPOSED, dW^T [K,N] = X^T . dY, so that the wide operand is the one that stays
    // token-major (half-TN form below) and only X is transposed; a transposing reduction writes dW.  The bias gradient, which the
    // transpose of dY used to carry, is a column-sum pass of its own when the caller wants it here (modcr_ffn_up_du_bwd has it from
    // the producing GEMM's epilogue): [3072 x 768] 591 -> 465 us without it, [2304 x 768] 499 -> 363 us + the pass.
    if (N > K && workspace && dy_dtype == MODCR_BF16 && dtype == MODCR_BF16 && modcr_knob_int("MODCR_GEMM_HALF_TN", 1) != 0 &&
        (N % 256) == 0 && (K % 256) == 0 && K >= 256 && (M % 64) == 0 && M >= 256 && (lddy % 8) == 0 && modcr_aligned16(dY) &&
        (int64_t)64 * lddy * 2 + (int64_t)N * 2 < (1ll << 31)) {
        const BwdWeightPlan ps = plan_bwd_weight(M, K, N);              // rows of the product = X features
        if (workspace_bytes >= ps.total && ps.kps >= 4 && !(ps.kps & 1) && (int64_t)K * ps.Mp < (1ll << 31)) {
            hipStream_t st = (hipStream_t)stream;
            bf16* xt = (bf16*)workspace;                                    // (the plan's dY^T slot: X^T [K, Mp] here)
            float* part = (float*)((char*)workspace + ps.off_part);
            LinearArgs p;
            p.A = xt; p.lda = ps.Mp; p.W = (const bf16*)dY; p.ldw = lddy; p.bias = nullptr; p.res = nullptr; p.ldr = 0;
            p.res_dtype = 0; p.C = part; p.ldc = N; p.out_dtype = MODCR_F32; p.M = K; p.N = N; p.K = (int)ps.Mp; p.kvalid = M;
            p.act = MODCR_ACT_NONE; p.tiles_m = p.tiles_n = 0; p.vec_ok = 1;
            p.k_tiles_per_split = ps.kps; p.split_stride = (int64_t)N * K;
            if (p8_ok(p)) {
                int rc = transpose_to_bf16(X, dtype, ldx, xt, ps.Mp, M, K, (int)ps.Mp, st);
                if (rc != MODCR_OK) return rc;
                rc = launch_p8d<MODCR_ACT_NONE, 0, MODCR_F32, 1, 2>(p, st);
                if (rc != MODCR_OK) return rc;
                hipLaunchKernelGGL(reduce_partials_transposed_kernel, dim3((unsigned)((N + 31) / 32), (unsigned)((K + 31) / 32)), dim3(256), 0, st,
                                   part, ps.splits, (int64_t)N * K, dW, K, N, accumulate);
                rc = modcr_check_launch("reduce_partials_transposed");
                if (rc != MODCR_OK || !db) return rc;
                // bias gradient: per-256-row-block column sums of dY (plain stores) + their fold in block order -- two launches as the
                // memset + atomic form it replaces, but reproducible; BEHIND the reduction: in front of the product, or between product and
                // reduction (where the fold could ride on the reduction's grid), its 425-566 MB read of dY costs more than it saves
                // (tools/ab_dw_bias.py: [2304 x 768] 421 -> 426 us; the reduction 14.6 -> 21.0 us with its partials evicted)
                if ((N % 8) == 0) {
                    float* dbp = (float*)((char*)workspace + ps.off_dbp);
                    const int nblk = (M + 255) / 256;
                    hipLaunchKernelGGL(colsum_bf16_block_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)nblk), dim3(256), 0, st,
                                       (const bf16*)dY, lddy, dbp, M, N);
                    rc = modcr_check_launch("colsum_bf16_block");
                    if (rc != MODCR_OK) return rc;
                    return launch_db_fold(DbFold{dbp, nblk, N, db, accumulate}, st);
                }
                hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, st, dY, lddy, (int)dy_dtype, db, M, N, accumulate);
                return modcr_check_launch("colsum");
            }
        }
    }
    const BwdWeightPlan pl = plan_bwd_weight(M, N, K);
    if (workspace && workspace_bytes >= pl.total) {
        hipStream_t st = (hipStream_t)stream;
        bf16* dyt = (bf16*)workspace;
        bf16* xt = (bf16*)((char*)workspace + pl.off_xt);
        float* part = (float*)((char*)workspace + pl.off_part);
        // db rides along with the transposition of dY (column sums of the tile the kernel holds) where the 16-byte path applies
        const bool fused_db = db && (N % 8) == 0 && (lddy % 8) == 0 && (pl.Mp % 64) == 0 && modcr_aligned16(dY) && modcr_aligned16(dyt);
        float* dbp = (float*)((char*)workspace + pl.off_dbp);
        int nblk = 0;
        int rc = transpose_to_bf16(dY, dy_dtype, lddy, dyt, pl.Mp, M, N, (int)pl.Mp, st, fused_db ? dbp : nullptr, &nblk);
        if (rc != MODCR_OK) return rc;
        const DbFold fold = fused_db ? DbFold{dbp, nblk, N, db, accumulate} : DbFold{nullptr, 0, 0, nullptr, 0};
        // half-TN form: X stays token-major, as the forward saved it (its transpose is the larger half of the transposed bytes of a
        // layer: x, ctx, a and the [M, 4H] FFN intermediate) -- the 256 x 256 kernel stages [64 tokens][128 features] images of it
        // and reads the fragments with ds_read_b64_tr_b16; dY^T is still transposed (the bias gradient rides on that pass and its
        // zero padding covers the tokens past M, for which the kernel re-reads X's last K-tile).  The transposed reads make the
        // product ~20 % slower (205 vs 170 us averaged over the encoder shapes at M = 92160), so the form is taken where X is at
        // least as wide as dY (K >= N: the transpose it saves is at least as large as the one that stays) -- dW of BertSelfOutput
        // 236 -> 190 us, of BertOutput 599 -> 470 us; [3072 x 768] (601 -> 644 us) and [2304 x 768] (490 -> 478 us) keep both
        // transposes (tools/run_bench_dw.sh).  MODCR_GEMM_HALF_TN = 0 / 2 (tuning build): never / whenever the shape allows.
        const int half_knob = modcr_knob_int("MODCR_GEMM_HALF_TN", 1);
        const bool half_tn = half_knob != 0 && (K >= N || half_knob == 2) && dtype == MODCR_BF16 && (N % 256) == 0 && N >= 256 && (K % 256) == 0 &&
                             (M % 64) == 0 && M >= 256 && (ldx % 8) == 0 && modcr_aligned16(X) && pl.kps >= 4 && !(pl.kps & 1) && pl.splits >= 1 &&
                             (int64_t)64 * ldx * 2 + (int64_t)K * 2 < (1ll << 31) && (int64_t)N * pl.Mp < (1ll << 31);
        LinearArgs ph;
        ph.A = dyt; ph.lda = pl.Mp; ph.W = (const bf16*)X; ph.ldw = ldx; ph.bias = nullptr; ph.res = nullptr; ph.ldr = 0;
        ph.res_dtype = 0; ph.C = part; ph.ldc = K; ph.out_dtype = MODCR_F32; ph.M = N; ph.N = K; ph.K = (int)pl.Mp; ph.kvalid = M;
        ph.act = MODCR_ACT_NONE; ph.tiles_m = ph.tiles_n = 0; ph.vec_ok = 1;
        ph.k_tiles_per_split = pl.kps; ph.split_stride = (int64_t)N * K;
        if (half_tn && p8_ok(ph)) {
            LinearArgs& p = ph;
            const bool direct_out = pl.splits == 1 && !accumulate;
            if (direct_out) p.C = dW;
            rc = launch_p8d<MODCR_ACT_NONE, 0, MODCR_F32, 1, 2>(p, st);
            if (rc != MODCR_OK) return rc;
            if (!direct_out) rc = launch_reduce_partials(part, pl.splits, (int64_t)N * K, dW, accumulate, fold, st);
            else if (fused_db) rc = launch_db_fold(fold, st);
            if (rc != MODCR_OK || !db || fused_db) return rc;
            hipLaunchKernelGGL(rowsum_bf16_kernel, dim3((N + 3) / 4), dim3(256), 0, st, dyt, pl.Mp, db, N, (int)pl.Mp, accumulate);
            return modcr_check_launch("rowsum");
        }
        rc = transpose_to_bf16(X, dtype, ldx, xt, pl.Mp, M, K, (int)pl.Mp, st);
        if (rc != MODCR_OK) return rc;
        LinearArgs p;
        p.A = dyt; p.lda = pl.Mp; p.W = xt; p.ldw = pl.Mp; p.bias = nullptr; p.res = nullptr; p.ldr = 0;
        p.res_dtype = 0; p.C = part; p.ldc = K; p.out_dtype = MODCR_F32; p.M = N; p.N = K; p.K = (int)pl.Mp;
        p.act = MODCR_ACT_NONE; p.tiles_m = p.tiles_n = 0;
        p.vec_ok = (K % 4 == 0); p.k_tiles_per_split = pl.kps; p.split_stride = (int64_t)N * K;
        const bool direct_out = pl.splits == 1 && !accumulate;      // one split: the product IS dW, no reduce pass
        if (direct_out) p.C = dW;
        rc = p8_ok(p) ? launch_p8<MODCR_ACT_NONE, 0, MODCR_F32>(p, st) : dispatch_linear(p, st);
        if (rc != MODCR_OK) return rc;
        if (!direct_out) rc = launch_reduce_partials(part, pl.splits, (int64_t)N * K, dW, accumulate, fold, st);
        else if (fused_db) rc = launch_db_fold(fold, st);
        if (rc != MODCR_OK || !db || fused_db) return rc;
        hipLaunchKernelGGL(rowsum_bf16_kernel, dim3((N + 3) / 4), dim3(256), 0, st, dyt, pl.Mp, db, N, (int)pl.Mp,
                           accumulate);
        return modcr_check_launch("rowsum");
    }
    // dW[n,k] = sum_m dY[m,n] X[m,k]
    GemmF32Args a;
    a.A = dY; a.sam = 1; a.sak = lddy; a.a_dtype = dy_dtype;
    a.B = X; a.sbk = ldx; a.sbn = 1; a.b_dtype = dtype;
    a.bias = nullptr; a.res = nullptr; a.ldr = 0; a.res_dtype = 0;
    a.C = dW; a.ldc = K; a.out_dtype = MODCR_F32; a.M = N; a.N = K; a.K = M; a.act = 0;
    a.accumulate = accumulate;
    int rc = launch_gemm_f32(a, (hipStream_t)stream);
    if (rc != MODCR_OK || !db) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)stream, dY,
                       lddy, (int)dy_dtype, db, M, N, accumulate);
    return modcr_check_launch("colsum");
}

// ---- LN(A.W^T + bias + residual): GEMM with fp32 pre-LN rows into the workspace, then row LN ----
extern "C" int modcr_linear_residual_ln_fwd(const void* A, int64_t lda, const void* W, const float* bias,
                                            const void* residual, const float* gamma, const float* beta,
                                            float eps, void* out, int32_t M, int32_t N, int32_t K,
                                            void* workspace, int64_t workspace_bytes, int32_t dtype,
                                            modcr_stream_t stream) {
    MODCR_REQUIRE(workspace && workspace_bytes >= (int64_t)M * N * 4,
                  "linear_residual_ln_fwd: workspace %lld < %lld bytes", (long long)workspace_bytes,
                  (long long)M * N * 4);
    // Opt-in experiment (MODCR_PRELN_BF16=1): only the sublayer's own output (A.W^T + bias) crosses HBM, as bf16,
    // and the LN pass adds the (bf16, exact) residual in fp32: the GEMM stores 2 bytes per element instead of 4
    // and does not read the residual.  Measured 33.1 vs 34.0 ms per step, but the extra rounding pushes the
    // 12-layer seq_enc pooled output to 0.066 against the 0.06 bound of tests/test_hip_models.py: off by default.
    const int preln_bf16 = modcr_knob_set("MODCR_PRELN_BF16");            // tuning build only
    if (dtype == MODCR_BF16 && preln_bf16) {
        int rc = modcr_linear_fwd(A, lda, W, K, bias, nullptr, 0, 0, workspace, N, M, N, K,
                                  MODCR_ACT_NONE, dtype, MODCR_BF16, stream);
        if (rc != MODCR_OK) return rc;
        return modcr_layernorm_fwd(workspace, MODCR_BF16, residual, MODCR_BF16, gamma, beta, eps, out, dtype, M, N, 0, 0, stream);
    }
    // bf16 path: the pre-LayerNorm rows (GEMM + bias + residual) cross HBM once as IEEE half -- 2 bytes like bf16 (which
    // measurably hurt: MODCR_PRELN_BF16 above) but with 11 significant bits, so the LayerNorm pass sees them 8x finer than the
    // bf16 rounding of its own output; the GEMM runs at its 2-byte-output rate (proj 86 -> 57 us, FFN-down 238 -> 205 us at
    // M = 46080) and the LayerNorm pass reads half the bytes.  fp32 parity path: fp32 rows.
    const int32_t pre_dt = dtype == MODCR_BF16 ? MODCR_F16 : MODCR_F32;
    int rc = modcr_linear_fwd(A, lda, W, K, bias, residual, N, dtype, workspace, N, M, N, K,
                              MODCR_ACT_NONE, dtype, pre_dt, stream);
    if (rc != MODCR_OK) return rc;
    return modcr_layernorm_fwd(workspace, pre_dt, nullptr, 0, gamma, beta, eps, out, dtype, M, N, 0, 0,
                               stream);
}

// ---- LN(dropout(A.W^T + bias) + residual): BertSelfOutput / BertOutput as ONE C-ABI call = two launches: the GEMM's own
// rows through the workspace (IEEE half on the bf16 path, fp32 on the parity path), then the row pass (mask, residual,
// LayerNorm).  (A single-kernel form was built and measured in round 2 -- a workgroup owning 192-row blocks, both 384-column
// halves of the 192 x 384 engine back to back, the first half parked as IEEE half in a lane-private slab, row statistics
// and normalisation in the second epilogue.  Correct, but 146 / 216 us (p = 0 / 0.3) against 97 us for the two launches at
// M = 46080, K = 768: with the accumulators filling the register file the epilogue spills, and every scratch reload waits,
// in the in-order vmcnt counter, behind the next pass's prologue DMAs.  Not kept.)
extern "C" int modcr_dropout_residual_ln_fwd(const void* x, int32_t x_dtype, const void* residual, int32_t res_dtype, const float* gamma,
                                             const float* beta, float eps, void* out, int32_t out_dtype, void* pre_out, int32_t pre_dtype,
                                             int64_t M, int32_t H, float p, uint64_t seed, uint64_t offset, modcr_stream_t stream);

extern "C" int64_t modcr_linear_dropout_residual_ln_workspace(int32_t M, int32_t N, int32_t K, int32_t dtype) {
    (void)K; (void)dtype;
    return (int64_t)M * N * 4;
}

extern "C" int modcr_linear_dropout_residual_ln_fwd(const void* A, int64_t lda, const void* W, const float* bias,
                                                    const void* residual, const float* gamma, const float* beta, float eps,
                                                    void* out, void* pre_out, int32_t pre_dtype, int32_t M, int32_t N, int32_t K, float p,
                                                    uint64_t seed, uint64_t offset, void* workspace, int64_t workspace_bytes,
                                                    int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(A && W && residual && gamma && beta && out && workspace, "linear_dropout_residual_ln_fwd: null pointer");
    MODCR_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K, "linear_dropout_residual_ln_fwd: bad shape");
    MODCR_REQUIRE(p >= 0.f && p < 1.f, "linear_dropout_residual_ln_fwd: p=%g out of [0, 1)", p);
    MODCR_REQUIRE(workspace_bytes >= modcr_linear_dropout_residual_ln_workspace(M, N, K, dtype), "linear_dropout_residual_ln_fwd: workspace too small");
    // two launches: the GEMM's own output (IEEE half on the bf16 path), then the row pass (mask when p > 0, residual, LayerNorm).
    // p = 0 takes the same two: with the residual added in the GEMM epilogue instead, the sublayer measured 206 / 410 us
    // (K = 768 / 3072, M = 92160) against 187 / 401 us this way (tools/ab_sublayer.py).
    // (a caller that wants the pre-LayerNorm rows -- a trainable layer -- gets them from fp32 GEMM rows: its backward
    // differentiates through them, and the half rounding of 24 layers' worth of rows showed in the G10 gradients.  Re-measured in
    // round 4: config 3 150.7 -> 146.8 ms per step with half rows everywhere, but G10's grad cls_ensemble.weight goes from 0.110 to
    // 0.179 relative L2 against a bound of 0.15 -- not adopted.)
    const int32_t pre_dt = (dtype == MODCR_BF16 && !pre_out) ? MODCR_F16 : MODCR_F32;
    int rc = modcr_linear_fwd(A, lda, W, K, bias, nullptr, 0, 0, workspace, N, M, N, K, MODCR_ACT_NONE, dtype, pre_dt, stream);
    if (rc != MODCR_OK) return rc;
    return modcr_dropout_residual_ln_fwd(workspace, pre_dt, residual, dtype, gamma, beta, eps, out, dtype, pre_out, pre_dtype, M, N, p, seed, offset, stream);
}

extern "C" int modcr_proj_residual_ln_fwd(const void* ctx, const void* wo, const float* bo, const void* x,
                                          const float* gamma, const float* beta, float eps, void* out,
                                          int32_t M, int32_t H, void* workspace, int64_t workspace_bytes,
                                          int32_t dtype, modcr_stream_t stream) {
    return modcr_linear_residual_ln_fwd(ctx, H, wo, bo, x, gamma, beta, eps, out, M, H, H, workspace,
                                        workspace_bytes, dtype, stream);
}

extern "C" int modcr_ffn_down_residual_ln_fwd(const void* inter, const void* w2, const float* b2,
                                              const void* a, const float* gamma, const float* beta,
                                              float eps, void* out, int32_t M, int32_t H, int32_t I,
                                              void* workspace, int64_t workspace_bytes, int32_t dtype,
                                              modcr_stream_t stream) {
    return modcr_linear_residual_ln_fwd(inter, I, w2, b2, a, gamma, beta, eps, out, M, H, I, workspace,
                                        workspace_bytes, dtype, stream);
}


// ---- backward composites of the encoder layer's three GEMM blocks (SURVEY 8b minimum ABI set) -------------------
static int64_t bwd_sub_ws(int32_t M, int32_t N, int32_t K) {
    const int64_t a = modcr_linear_bwd_input_workspace(M, N, K), b = modcr_linear_bwd_weight_workspace(M, N, K);
    return ((a > b ? a : b) + 255) & ~(int64_t)255;
}
static int64_t dsub_bytes(int32_t M, int32_t N) { return (((int64_t)M * N * 2) + 255) & ~(int64_t)255; }
extern "C" int64_t modcr_linear_residual_ln_bwd_workspace(int32_t M, int32_t N, int32_t K) { return bwd_sub_ws(M, N, K) + dsub_bytes(M, N); }

extern "C" int modcr_layernorm_dropout_bwd(const void* dY, int32_t dy_dtype, const void* pre, int32_t pre_dtype, const float* gamma, float eps,
                                           float* d_pre, void* d_sub_bf16, float* dgamma, float* dbeta, int64_t M, int32_t H,
                                           float p, uint64_t seed, uint64_t offset, modcr_stream_t stream);

extern "C" int modcr_linear_residual_ln_dropout_bwd(const void* dY, int32_t dy_dtype, const void* pre, int32_t pre_dtype, const void* A, int64_t lda,
                                                    const void* W, const float* gamma, float eps, float* d_pre, void* dA, float* dW,
                                                    float* dbias, float* dgamma, float* dbeta, int32_t M, int32_t N, int32_t K,
                                                    float p, uint64_t seed, uint64_t offset, void* workspace,
                                                    int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(dY && pre && A && W && gamma && d_pre && dA && dW && dbias && dgamma && dbeta, "linear_residual_ln_bwd: null pointer");
    MODCR_REQUIRE(lda == K, "linear_residual_ln_bwd: A must be dense (lda == K)");
    const bool mfma = dtype == MODCR_BF16;
    MODCR_REQUIRE(!mfma || (workspace && workspace_bytes >= modcr_linear_residual_ln_bwd_workspace(M, N, K)),
                  "linear_residual_ln_bwd: workspace too small");
    if (mfma && (N % 256) == 0 && N <= 1024) {
        // bf16 route: one LayerNorm-backward pass leaves d_pre (fp32, residual branch) and the GEMMs' operand: a bf16
        // copy with the forward's dropout mask applied (no separate mask / cast passes, bf16 transposes)
        void* dsub = workspace;
        void* sub = (char*)workspace + dsub_bytes(M, N);
        const int64_t sub_bytes = workspace_bytes - dsub_bytes(M, N);
        int rc = modcr_layernorm_dropout_bwd(dY, dy_dtype, pre, pre_dtype, gamma, eps, d_pre, dsub, dgamma, dbeta, M, N, p, seed, offset, stream);
        if (rc != MODCR_OK) return rc;
        rc = modcr_linear_bwd_weight(dsub, N, MODCR_BF16, A, K, dW, dbias, M, N, K, 0, dtype, sub, sub_bytes, stream);
        if (rc != MODCR_OK) return rc;
        return modcr_linear_bwd_input(dsub, N, MODCR_BF16, W, K, dA, K, M, N, K, dtype, dtype, sub, sub_bytes, stream);
    }
    MODCR_REQUIRE(p == 0.f && dy_dtype == MODCR_F32 && pre_dtype == MODCR_F32, "linear_residual_ln_bwd: dropout / bf16 dY / half pre-LN rows need the bf16 route with N in {256, 512, 768, 1024}");
    void* sub = mfma ? workspace : nullptr;
    const int64_t sub_bytes = mfma ? workspace_bytes : 0;
    // LayerNorm over the saved pre-LN rows: d_pre is the gradient of the GEMM output AND of the residual
    int rc = modcr_layernorm_bwd((const float*)dY, (const float*)pre, nullptr, gamma, eps, d_pre, dgamma, dbeta, M, N, stream);
    if (rc != MODCR_OK) return rc;
    rc = modcr_linear_bwd_weight(d_pre, N, MODCR_F32, A, K, dW, dbias, M, N, K, 0, dtype, sub, sub_bytes, stream);
    if (rc != MODCR_OK) return rc;
    return modcr_linear_bwd_input(d_pre, N, MODCR_F32, W, K, dA, K, M, N, K, dtype, dtype, sub, sub_bytes, stream);
}

extern "C" int modcr_linear_residual_ln_bwd(const float* dY, const float* pre, const void* A, int64_t lda, const void* W,
                                            const float* gamma, float eps, float* d_pre, void* dA, float* dW,
                                            float* dbias, float* dgamma, float* dbeta, int32_t M, int32_t N, int32_t K,
                                            void* workspace, int64_t workspace_bytes, int32_t dtype,
                                            modcr_stream_t stream) {
    return modcr_linear_residual_ln_dropout_bwd(dY, MODCR_F32, pre, MODCR_F32, A, lda, W, gamma, eps, d_pre, dA, dW, dbias, dgamma, dbeta,
                                                M, N, K, 0.f, 0, 0, workspace, workspace_bytes, dtype, stream);
}

extern "C" int modcr_proj_residual_ln_bwd(const float* dY, const float* pre, const void* ctx, const void* wo,
                                          const float* gamma, float eps, float* d_pre, void* dctx, float* dwo,
                                          float* dbo, float* dgamma, float* dbeta, int32_t M, int32_t H,
                                          void* workspace, int64_t workspace_bytes, int32_t dtype,
                                          modcr_stream_t stream) {
    return modcr_linear_residual_ln_bwd(dY, pre, ctx, H, wo, gamma, eps, d_pre, dctx, dwo, dbo, dgamma, dbeta, M, H, H,
                                        workspace, workspace_bytes, dtype, stream);
}

extern "C" int modcr_ffn_down_residual_ln_bwd(const float* dY, const float* pre, const void* inter, const void* w2,
                                              const float* gamma, float eps, float* d_pre, void* dinter, float* dw2,
                                              float* db2, float* dgamma, float* dbeta, int32_t M, int32_t H, int32_t I,
                                              void* workspace, int64_t workspace_bytes, int32_t dtype,
                                              modcr_stream_t stream) {
    return modcr_linear_residual_ln_bwd(dY, pre, inter, I, w2, gamma, eps, d_pre, dinter, dw2, db2, dgamma, dbeta, M, H, I,
                                        workspace, workspace_bytes, dtype, stream);
}

// ---- trainable FFN with the GELU input kept (bf16 route): the forward's FFN-up writes u = a.W1^T + b1 beside gelu(u), the
// FFN-down dX product multiplies by gelu'(u) in its epilogue and hands back d_u, and the FFN-up backward is then two products
// (dW1, dX) instead of three: the recompute GEMM of modcr_ffn_up_gelu_bwd (0.48 ms per layer at M = 92160) is gone for
// +1 row of bf16 stores in the forward.  Shapes outside the persistent kernel's (small M, ragged H / I) keep the recompute route:
// modcr_ffn_keep_supported says which.
static void ffn_keep_args(LinearArgs& p, const void* x, const void* w1, const float* b1, void* out, void* pre_act, int32_t M, int32_t H, int32_t I) {
    p.A = (const bf16*)x; p.lda = H; p.W = (const bf16*)w1; p.ldw = H; p.bias = b1;
    p.res = nullptr; p.ldr = 0; p.res_dtype = 0; p.C = out; p.ldc = I; p.C2 = pre_act;
    p.out_dtype = MODCR_BF16; p.M = M; p.N = I; p.K = H; p.act = MODCR_ACT_GELU_KEEP;
    p.tiles_m = p.tiles_n = 0; p.vec_ok = 1; p.k_tiles_per_split = 0; p.split_stride = 0;
}
extern "C" int modcr_ffn_keep_supported(int32_t M, int32_t H, int32_t I, int32_t dtype) {
    if (dtype != MODCR_BF16 || M < 256 || (M % 8) != 0 || (H % 256) != 0 || H > 1024 || (I % 256) != 0) return 0;
    if ((int64_t)M * I >= (1ll << 30)) return 0;                // 32-bit byte offsets of the dX product's A operand
    return 1;
}

extern "C" int modcr_ffn_up_gelu_keep_fwd(const void* x, const void* w1, const float* b1, void* out, void* pre_act,
                                          int32_t M, int32_t H, int32_t I, int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(x && w1 && b1 && out && pre_act, "ffn_up_gelu_keep_fwd: null pointer");
    MODCR_REQUIRE(modcr_ffn_keep_supported(M, H, I, dtype), "ffn_up_gelu_keep_fwd: shape / dtype outside the kept-input route (see modcr_ffn_keep_supported)");
    LinearArgs p;
    ffn_keep_args(p, x, w1, b1, out, pre_act, M, H, I);
    MODCR_REQUIRE(modcr_aligned16(x) && modcr_aligned16(w1) && modcr_aligned16(b1) && modcr_aligned16(pre_act) && p8_ok(p),
                  "ffn_up_gelu_keep_fwd: operands must be 16-byte aligned");
    // whole tiles: the seamless-ring kernel (linear_bf16_p8_kernel, SPEC)
    if ((M % 256) == 0 && modcr_knob_int("MODCR_GEMM_SPEC", 1)) return launch_p8d<MODCR_ACT_GELU_KEEP, 0, MODCR_BF16, 1, 0, 1>(p, (hipStream_t)stream);
    return launch_p8d<MODCR_ACT_GELU_KEEP, 0, MODCR_BF16, 1>(p, (hipStream_t)stream);
}

extern "C" int64_t modcr_ffn_down_gelu_bwd_workspace(int32_t M, int32_t H, int32_t I) { return modcr_linear_residual_ln_bwd_workspace(M, H, I); }

extern "C" int modcr_ffn_down_residual_ln_gelu_bwd(const void* dY, int32_t dy_dtype, const void* pre, int32_t pre_dtype, const void* inter, const void* w2,
                                                   const float* gamma, float eps, const void* pre_act, float* d_pre, void* d_u, float* db_u,
                                                   float* dw2, float* db2, float* dgamma, float* dbeta, int32_t M, int32_t H, int32_t I,
                                                   float p, uint64_t seed, uint64_t offset, void* workspace, int64_t workspace_bytes,
                                                   int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(dY && pre && inter && w2 && gamma && pre_act && d_pre && d_u && dw2 && db2 && dgamma && dbeta, "ffn_down_gelu_bwd: null pointer");
    MODCR_REQUIRE(modcr_ffn_keep_supported(M, H, I, dtype), "ffn_down_gelu_bwd: shape / dtype outside the kept-input route (see modcr_ffn_keep_supported)");
    MODCR_REQUIRE(workspace && workspace_bytes >= modcr_ffn_down_gelu_bwd_workspace(M, H, I), "ffn_down_gelu_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    void* dsub = workspace;
    void* sub = (char*)workspace + dsub_bytes(M, H);
    const int64_t sub_bytes = workspace_bytes - dsub_bytes(M, H);
    int rc = modcr_layernorm_dropout_bwd(dY, dy_dtype, pre, pre_dtype, gamma, eps, d_pre, dsub, dgamma, dbeta, M, H, p, seed, offset, stream);
    if (rc != MODCR_OK) return rc;
    rc = modcr_linear_bwd_weight(dsub, H, MODCR_BF16, inter, I, dw2, db2, M, H, I, 0, dtype, sub, sub_bytes, stream);
    if (rc != MODCR_OK) return rc;
    // d_u[M,I] = (d_sub[M,H] . W2[H,I]) * gelu'(u): the NT product of d_sub with W2^T [I,H], the saved GELU input as the
    // multiplicative operand of the epilogue
    bf16* wt = (bf16*)sub;
    MODCR_REQUIRE(sub_bytes >= (int64_t)I * H * 2, "ffn_down_gelu_bwd: workspace too small");
    rc = transpose_to_bf16(w2, dtype, I, wt, H, H, I, H, st);
    if (rc != MODCR_OK) return rc;
    LinearArgs a;
    a.A = (const bf16*)dsub; a.lda = H; a.W = wt; a.ldw = H; a.bias = nullptr;
    a.res = pre_act; a.ldr = I; a.res_dtype = MODCR_BF16; a.C = d_u; a.ldc = I;
    a.out_dtype = MODCR_BF16; a.M = M; a.N = I; a.K = H; a.act = MODCR_ACT_MUL_GELU_GRAD;
    a.tiles_m = a.tiles_n = 0; a.vec_ok = 1; a.k_tiles_per_split = 0; a.split_stride = 0;
    MODCR_REQUIRE(modcr_aligned16(pre_act) && modcr_aligned16(d_u) && p8_ok(a), "ffn_down_gelu_bwd: operands must be 16-byte aligned");
    if (db_u) {          // column sums of d_u (= the bias gradient of BertIntermediate) from the same epilogue
        if (hipMemsetAsync(db_u, 0, (size_t)I * sizeof(float), st) != hipSuccess) {
            modcr_set_error("ffn_down_gelu_bwd: hipMemsetAsync failed");
            return MODCR_ERR_LAUNCH;
        }
        a.C2 = db_u;
    }
    return launch_p8d<MODCR_ACT_MUL_GELU_GRAD, 1, MODCR_BF16, 0>(a, st);
}

extern "C" int64_t modcr_ffn_up_du_bwd_workspace(int32_t M, int32_t H, int32_t I) { return bwd_sub_ws(M, I, H); }

// FFN-up backward from d_u (the gradient of the GELU input): dW1 = d_u^T.x, db1 = colsum(d_u), dx = d_u.W1 (+ dx_residual)
extern "C" int modcr_ffn_up_du_bwd(const void* du, const void* x, const void* w1, const float* dx_residual, void* dx, float* dw1,
                                   float* db1, int32_t M, int32_t H, int32_t I, void* workspace, int64_t workspace_bytes,
                                   int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(du && x && w1 && dx && dw1, "ffn_up_du_bwd: null pointer");          // db1 may be NULL: the caller has it (db_u of modcr_ffn_down_residual_ln_gelu_bwd)
    MODCR_REQUIRE(modcr_ffn_keep_supported(M, H, I, dtype), "ffn_up_du_bwd: shape / dtype outside the kept-input route (see modcr_ffn_keep_supported)");
    MODCR_REQUIRE(workspace && workspace_bytes >= modcr_ffn_up_du_bwd_workspace(M, H, I), "ffn_up_du_bwd: workspace too small");
    int rc = modcr_linear_bwd_weight(du, I, MODCR_BF16, x, H, dw1, db1, M, I, H, 0, dtype, workspace, workspace_bytes, stream);
    if (rc != MODCR_OK) return rc;
    // dx (the gradient of BertSelfOutput's output = the dY of its LayerNorm backward) leaves in the storage dtype: the sum with the residual
    // branch is formed in fp32 in the epilogue and rounded once, as the gradient that crosses a layer boundary is (round 4: the fp32 copy cost
    // 141 MB more in the GEMM's store and in the LayerNorm backward's read: config 3 150.9 -> 150.0 ms, tolerance uses unchanged)
    return modcr_linear_bwd_input_res(du, I, MODCR_BF16, w1, H, dx_residual, H, dx, H, M, I, H, dtype, dtype, workspace, workspace_bytes, stream);
}

extern "C" int64_t modcr_ffn_up_gelu_bwd_workspace(int32_t M, int32_t H, int32_t I) {
    const int64_t rows = (((int64_t)M * I * (int64_t)sizeof(float)) + 255) & ~(int64_t)255;
    return 2 * rows + bwd_sub_ws(M, I, H);
}

extern "C" int modcr_ffn_up_gelu_bwd(const void* dinter, int32_t dinter_dtype, const void* x, const void* w1, const float* b1,
                                     const float* dx_residual, float* dx, float* dw1, float* db1, int32_t M, int32_t H, int32_t I,
                                     void* workspace, int64_t workspace_bytes, int32_t dtype, modcr_stream_t stream) {
    MODCR_REQUIRE(dinter && x && w1 && b1 && dx && dw1 && db1, "ffn_up_gelu_bwd: null pointer");
    MODCR_REQUIRE(workspace && workspace_bytes >= modcr_ffn_up_gelu_bwd_workspace(M, H, I), "ffn_up_gelu_bwd: workspace too small");
    const int64_t rows = (((int64_t)M * I * (int64_t)sizeof(float)) + 255) & ~(int64_t)255;
    float* u = (float*)workspace;
    float* du = (float*)((char*)workspace + rows);
    const bool mfma = dtype == MODCR_BF16;
    void* sub = mfma ? (void*)((char*)workspace + 2 * rows) : nullptr;
    const int64_t sub_bytes = mfma ? bwd_sub_ws(M, I, H) : 0;
    // the GELU input is recomputed (one GEMM) instead of saved
    if (mfma && dinter_dtype == MODCR_BF16) {
        // bf16 route: d_u = gelu'(x.W1^T + b1) * d_inter inside the recompute GEMM's epilogue (d_inter rides in as the
        // residual operand, multiplied instead of added), written once as bf16: no fp32 pre-activation round trip,
        // no separate activation-backward or cast pass
        LinearArgs p;
        p.A = (const bf16*)x; p.lda = H; p.W = (const bf16*)w1; p.ldw = H; p.bias = b1;
        p.res = dinter; p.ldr = I; p.res_dtype = MODCR_BF16; p.C = u; p.ldc = I;
        p.out_dtype = MODCR_BF16; p.M = M; p.N = I; p.K = H; p.act = MODCR_ACT_GELU_GRAD;
        p.tiles_m = p.tiles_n = 0; p.vec_ok = 1; p.k_tiles_per_split = 0; p.split_stride = 0;
        if ((H % 64) == 0 && modcr_aligned16(x) && modcr_aligned16(w1) && modcr_aligned16(dinter) && p8_ok(p)) {
            int rc = launch_p8d<MODCR_ACT_GELU_GRAD, 1, MODCR_BF16, 0>(p, (hipStream_t)stream);
            if (rc != MODCR_OK) return rc;
            rc = modcr_linear_bwd_weight(u, I, MODCR_BF16, x, H, dw1, db1, M, I, H, 0, dtype, sub, sub_bytes, stream);
            if (rc != MODCR_OK) return rc;
            return modcr_linear_bwd_input_res(u, I, MODCR_BF16, w1, H, dx_residual, H, dx, H, M, I, H, dtype, MODCR_F32, sub, sub_bytes, stream);
        }
    }
    int rc = modcr_linear_fwd(x, H, w1, H, b1, nullptr, 0, 0, u, I, M, I, H, MODCR_ACT_NONE, dtype, MODCR_F32, stream);
    if (rc != MODCR_OK) return rc;
    const float* dact = (const float*)dinter;
    if (dinter_dtype == MODCR_BF16) {
        rc = modcr_convert(dinter, MODCR_BF16, du, MODCR_F32, (int64_t)M * I, stream);
        if (rc != MODCR_OK) return rc;
        dact = du;
    }
    rc = modcr_act_bwd(dact, u, du, (int64_t)M * I, MODCR_ACT_GELU, stream);
    if (rc != MODCR_OK) return rc;
    rc = modcr_linear_bwd_weight(du, I, MODCR_F32, x, H, dw1, db1, M, I, H, 0, dtype, sub, sub_bytes, stream);
    if (rc != MODCR_OK) return rc;
    return modcr_linear_bwd_input_res(du, I, MODCR_F32, w1, H, dx_residual, H, dx, H, M, I, H, dtype, MODCR_F32, sub, sub_bytes, stream);
}
