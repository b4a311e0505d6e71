// Which CU-side path moves L2-resident operand bytes fastest on gfx950?  One 512-thread workgroup per CU re-reads a private 256-KiB
// region (L2-resident after the first pass) N times through: (1) global_load_lds dwordx4 (LDS-DMA), 8 rows x 128 B per instruction,
// (2) the same with 1 KiB contiguous per instruction, (3) global_load_dwordx4 into VGPRs (no LDS), (4) global_load_dwordx4 ->
// VGPR -> ds_write_b128 (register staging), (5) LDS-DMA issued by ONE wave per SIMD pair ... Prints GB/s per CU and B/clk at 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_feed_probe l2_feed_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#ifndef REGION
#define REGION (64 * 1024)
#endif
#define NBLK (REGION / 1024)
template <int MODE>
__global__ void __launch_bounds__(512, 2) probe(const char* __restrict__ src, int iters, u32x4* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const char* base = src + (size_t)blockIdx.x * REGION;
    u32x4 accv = {0, 0, 0, 0};
    // region = 256 "rows" of 1 KiB; one pass = 256 KiB = 32 instructions per wave
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 2 || MODE == 5) {
            const int nw = MODE == 5 ? 4 : 8;
            if (MODE == 5 && wid >= 4) { continue; }
#pragma unroll 4
            for (int j = 0; j < NBLK / nw; ++j) {
                const int blk = j * nw + (wid % nw);                // 1-KiB block
                size_t off;
                if (MODE == 2) off = (size_t)blk * 1024 + lane * 16;                           // contiguous KiB
                else off = (size_t)((blk & 7) * 8 + (lane >> 3)) * 1024 + (blk >> 3) * 128 + (lane & 7) * 16;   // 8 rows x 128 B, 1-KiB row pitch (64 rows x 1 KiB region)
                char* dst = smem + ((j & 15) * 8 + wid) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                if ((j & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
        } else if (MODE == 3 || MODE == 4) {
#pragma unroll 1
            for (int j0 = 0; j0 < NBLK / 8; j0 += 8) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int blk = (j0 + j) * 8 + wid;
                    const size_t off = (size_t)((blk & 7) * 8 + (lane >> 3)) * 1024 + (blk >> 3) * 128 + (lane & 7) * 16;
                    v[j] = *(const u32x4*)(base + off);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (MODE == 4) *(u32x4*)(smem + (((j0 + j) & 15) * 8 + wid) * 1024 + lane * 16) = v[j];
                    else accv ^= v[j];
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 4 || MODE == 1 || MODE == 2 || MODE == 5) { __syncthreads(); accv = *(u32x4*)(smem + tid * 16); }
    if (accv[0] == 0x12345678u) sink[blockIdx.x * 512 + tid] = accv;
}

template <int MODE>
static void run(const char* name, const char* src, u32x4* sink, int ncu, int iters) {
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<ncu, 512, 131072>>>(src, 2, sink);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        probe<MODE><<<ncu, 512, 131072>>>(src, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)ncu * REGION * iters;
    printf("%-44s %8.3f ms  %7.2f TB/s chip  %6.1f GB/s/CU  %5.1f B/clk/CU@2.4GHz\n", name, best, bytes / best / 1e9, bytes / best / 1e6 / ncu, bytes / best / 1e6 / ncu / 2.4);
}

int main() {
    int ncu = 256;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); ncu = prop.multiProcessorCount;
    char* src; u32x4* sink;
    hipMalloc(&src, (size_t)ncu * REGION); hipMemset(src, 1, (size_t)ncu * REGION);
    hipMalloc(&sink, (size_t)ncu * 512 * 16);
    const int iters = 800;
    run<1>("LDS-DMA dwordx4, 8 rows x 128 B / instr", src, sink, ncu, iters);
    run<2>("LDS-DMA dwordx4, 1 KiB contiguous / instr", src, sink, ncu, iters);
    run<5>("LDS-DMA, issued by 4 of 8 waves", src, sink, ncu, iters);
    run<3>("global_load_dwordx4 -> VGPR", src, sink, ncu, iters);
    run<4>("global_load_dwordx4 -> VGPR -> ds_write_b128", src, sink, ncu, iters);
    return 0;
}
