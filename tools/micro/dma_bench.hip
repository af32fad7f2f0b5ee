// Micro-benchmark: per-CU LDS-DMA (global_load_lds_dwordx4) fill rate for the GEMM staging pattern
// (1-KiB pieces = 8 rows x 128 B at a row stride), as a function of workgroup size, pieces in flight
// and where the rows live (L2-resident vs HBM).  hipcc --offload-arch=gfx950 -O3 dma_bench.hip -o dma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// each workgroup streams `tiles` tiles of `chunks_per_wave * nwaves` KiB; DEPTH tiles in flight
template <int DEPTH>
__global__ void dma_kernel(const char* src, size_t row_stride, size_t span_rows, int tiles, int cpw, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const size_t tile_bytes = (size_t)cpw * nw * 1024;
    // rows of this workgroup: a private window of the source so workgroups do not share lines
    const size_t row0 = ((size_t)blockIdx.x * 977) % span_rows;
    int issued = 0;
    auto issue = [&](int t) {
        for (int q = 0; q < cpw; ++q) {
            const int chunk = wave + q * nw;
            const size_t row = (row0 + (size_t)chunk * 8 + (lane >> 3)) % span_rows;
            const char* g = src + row * row_stride + (size_t)(t % 48) * 128 + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(smem + (t % DEPTH) * tile_bytes + chunk * 1024), 16, 0, 0);
        }
    };
    for (int t = 0; t < DEPTH - 1 && t < tiles; ++t) { issue(t); ++issued; }
    float acc = 0.f;
    for (int t = 0; t < tiles; ++t) {
        if (issued < tiles) { issue(issued); ++issued; }
        // wait for tile t: leave min(DEPTH-1, issued-1-t) tiles outstanding (approximate with full drain at the tail)
        if (issued - 1 - t >= DEPTH - 1 && DEPTH > 1) {
            if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8) : "memory");
            if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16) : "memory");
            if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(24) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        acc += *reinterpret_cast<float*>(smem + (t % DEPTH) * tile_bytes + threadIdx.x * 4);
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 123.456f) sink[0] = acc;
}

// register-staged variant: global_load_dwordx4 -> registers -> ds_write_b128, one tile ahead
__global__ void reg_kernel(const char* src, size_t row_stride, size_t span_rows, int tiles, int cpw_unused, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const size_t row0 = ((size_t)blockIdx.x * 977) % span_rows;
    uint4 rg[8];
    auto issue = [&](int t) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int chunk = wave + q * nw;
            const size_t row = (row0 + (size_t)chunk * 8 + (lane >> 3)) % span_rows;
            rg[q] = *reinterpret_cast<const uint4*>(src + row * row_stride + (size_t)(t % 48) * 128 + (lane & 7) * 16);
        }
    };
    float acc = 0.f;
    issue(0);
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            *reinterpret_cast<uint4*>(smem + (wave + q * nw) * 1024 + lane * 16) = rg[q];
        if (t + 1 < tiles) issue(t + 1);
        __syncthreads();
        acc += *reinterpret_cast<float*>(smem + threadIdx.x * 4);
        __syncthreads();
    }
    if (acc == 123.456f) sink[0] = acc;
}

float run_reg(const char* src, size_t row_stride, size_t span_rows, int tiles, int nwaves, int blocks, float* sink) {
    const size_t smem = (size_t)8 * nwaves * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(reg_kernel, dim3(blocks), dim3(nwaves * 64), smem, 0, src, row_stride, span_rows, tiles, 8, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(reg_kernel, dim3(blocks), dim3(nwaves * 64), smem, 0, src, row_stride, span_rows, tiles, 8, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 5.0 * blocks * (double)tiles * 8 * nwaves * 1024;
    return (float)(bytes / (ms * 1e-3) / 1e9);
}

template <int DEPTH>
float run(const char* src, size_t row_stride, size_t span_rows, int tiles, int nwaves, int blocks, float* sink) {
    const int cpw = 8;      // vmcnt immediates above assume 8 pieces per wave per tile
    const size_t smem = (size_t)DEPTH * cpw * nwaves * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(blocks), dim3(nwaves * 64), smem, 0, src, row_stride, span_rows, tiles, cpw, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(blocks), dim3(nwaves * 64), smem, 0, src, row_stride, span_rows, tiles, cpw, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 5.0 * blocks * (double)tiles * cpw * nwaves * 1024;
    return (float)(bytes / (ms * 1e-3) / 1e9);     // GB/s chip-wide
}

int main() {
    const size_t row_stride = 6144;                 // K = 3072 bf16
    char* big; float* sink;
    const size_t big_rows = 262144;                 // 1.5 GiB: HBM
    hipMalloc(&big, big_rows * row_stride); hipMemset(big, 1, big_rows * row_stride);
    hipMalloc(&sink, 64);
    printf("pattern: 1-KiB pieces = 8 rows x 128 B, row stride %zu B, 8 pieces/wave/tile, 256 workgroups (1/CU), 200 tiles each\n", row_stride);
    for (int nw : {4, 8, 12, 16}) {
        if (nw > 8) {
            for (size_t span : {(size_t)2048, big_rows}) {
                float a = run<1>(big, row_stride, span, 200, nw, 256, sink);
                float rgv = run_reg(big, row_stride, span, 200, nw, 256, sink);
                printf("waves=%d span=%zu rows: LDS-DMA depth1 %7.0f GB/s (%5.1f/CU)  register-staged %7.0f (%5.1f/CU)\n", nw, span, a, a / 256, rgv, rgv / 256);
            }
            continue;
        }
        for (size_t span : {(size_t)2048, (size_t)65536, big_rows}) {       // 12 MB (L2/MALL), 384 MB, 1.5 GB
            float a = run<1>(big, row_stride, span, 200, nw, 256, sink);
            float b = run<2>(big, row_stride, span, 200, nw, 256, sink);
            float c = nw == 4 ? run<3>(big, row_stride, span, 200, nw, 256, sink) : 0.f;
            float d = nw == 4 ? run<4>(big, row_stride, span, 200, nw, 256, sink) : 0.f;
            float rgv = run_reg(big, row_stride, span, 200, nw, 256, sink);
            printf("   register-staged (1 tile ahead): %7.0f GB/s chip, %5.1f per CU\n", rgv, rgv / 256);
            printf("waves=%d tile=%2d KiB span=%6zu rows (%7.1f MB): depth1 %7.0f  depth2 %7.0f  depth3 %7.0f  depth4 %7.0f GB/s chip  (per CU: %5.1f %5.1f %5.1f %5.1f)\n",
                   nw, 8 * nw, span, span * row_stride / 1e6, a, b, c, d, a / 256, b / 256, c / 256, d / 256);
        }
    }
    return 0;
}
