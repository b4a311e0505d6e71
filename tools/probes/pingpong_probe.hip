// What stretches a ping-pong tick?  8-wave workgroup per CU, two groups one barrier apart, each phase = [load segment] barrier
// [8 MFMAs] barrier.  The load segment is built up piece by piece: NR fragment reads (ds_read_b128 or ds_read_b64_tr_b16 pairs),
// ND LDS-DMA instructions per wave (from a small L2-resident buffer), a counted vmcnt.  Prints shader cycles per phase (512 = MFMA-bound).
// Build: hipcc --offload-arch=gfx950 -O3 -o pingpong_probe pingpong_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_hw;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// RD: 0 none, 1 = 6 x ds_read_b128, 2 = 12 x ds_read_b128, 3 = 12 x ds_read_b64_tr_b16;  ND = LDS-DMA instructions per wave per phase; PRIO = s_setprio around MFMAs
template <int RD, int ND, int PRIO, int WAITPOS>
__global__ void __launch_bounds__(512, 2) probe(const char* __restrict__ src, int iters, long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)tid;
    u32x4 fr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) fr[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f003f00u, 0x3f003f00u};
    const char* base = src + (size_t)blockIdx.x * 65536;
    const uint32_t rdoff = (uint32_t)(((lane & 31) * 128 + (((lane >> 5) ^ ((lane >> 1) & 7)) * 16)) + (wid & 3) * 4096);
    const uint32_t troff = (uint32_t)((lane & 15) * 512 + (lane >> 4) * 64 + (wid & 3) * 16);
    if (wid >= 4) __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const uint32_t sb = (uint32_t)(it & 7) * 16384u;
        if (RD == 1 || RD == 2) {
#pragma unroll
            for (int i = 0; i < (RD == 1 ? 6 : 12); ++i) {
                u32x4 t;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(sb + rdoff), "n"((i % 6) * 1024 * 2) : "memory");
                if (i < 6) fr[i] = t; else fr[i - 6] ^= t;
            }
        } else if (RD == 3) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                u32x2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(sb + troff), "n"(i * 64), "n"(i * 64 + 2048) : "memory");
                fr[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const size_t off = (size_t)(((it * ND + d) & 7) * 8 + (lane >> 3)) * 1024 + wid * 128 + (lane & 7) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                             (__attribute__((address_space(3))) void*)(smem + (((it + 6) & 7) * 16 + wid * 2 + d) * 1024), 16, 0, 0);
        }
        if (ND > 0 && WAITPOS == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * ND) : "memory");
        if (PRIO) asm volatile("s_barrier\n\ts_setprio 1\n\ts_waitcnt lgkmcnt(0)" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]), "+v"(fr[4]), "+v"(fr[5])::"memory");
        else asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]), "+v"(fr[4]), "+v"(fr[5])::"memory");
#pragma unroll
        for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, fr[i % 4]), __builtin_bit_cast(bf16x8_hw, fr[4 + (i & 1)]), acc[i], 0, 0, 0);
        if (ND > 0 && WAITPOS == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * ND) : "memory");
        if (PRIO) asm volatile("s_setprio 0\n\ts_barrier" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])::"memory");
        else asm volatile("s_barrier" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])::"memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wid < 4) __builtin_amdgcn_s_barrier();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) sink[tid] = s;
    if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int RD, int ND, int PRIO, int WAITPOS>
static void run(const char* name, const char* src, long long* out, float* sink, int ncu, int iters) {
    (void)hipFuncSetAttribute((const void*)probe<RD, ND, PRIO, WAITPOS>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<RD, ND, PRIO, WAITPOS><<<ncu, 512, 131072>>>(src, 10, out, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<RD, ND, PRIO, WAITPOS><<<ncu, 512, 131072>>>(src, iters, out, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    printf("%-66s %7.1f cycles/phase (512 = MFMA-bound)  %6.1f ns/phase  -> %.2f GHz\n", name, (double)c / iters, ms * 1e6 / iters, (double)c / iters / (ms * 1e6 / iters));
}

int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    char* src; long long* out; float* sink;
    (void)hipMalloc(&src, (size_t)ncu * 65536); (void)hipMemset(src, 0x3f, (size_t)ncu * 65536);
    (void)hipMalloc(&out, ncu * 8); (void)hipMalloc(&sink, 4096);
    const int iters = 20000;
    run<0, 0, 1, 0>("MFMA only, setprio", src, out, sink, ncu, iters);
    run<1, 0, 1, 0>("+ 6 ds_read_b128", src, out, sink, ncu, iters);
    run<2, 0, 1, 0>("+ 12 ds_read_b128", src, out, sink, ncu, iters);
    run<3, 0, 1, 0>("+ 12 ds_read_b64_tr_b16", src, out, sink, ncu, iters);
    run<3, 0, 0, 0>("+ 12 ds_read_b64_tr_b16, no setprio", src, out, sink, ncu, iters);
    run<0, 2, 1, 0>("MFMA + 2 LDS-DMA/wave (L2), vmcnt(10) in load segment", src, out, sink, ncu, iters);
    run<0, 2, 1, 1>("MFMA + 2 LDS-DMA/wave (L2), vmcnt(10) after MFMAs", src, out, sink, ncu, iters);
    run<1, 2, 1, 0>("6 b128 + 2 LDS-DMA, wait in load segment", src, out, sink, ncu, iters);
    run<1, 2, 1, 1>("6 b128 + 2 LDS-DMA, wait after MFMAs", src, out, sink, ncu, iters);
    run<3, 2, 1, 0>("12 tr + 2 LDS-DMA, wait in load segment", src, out, sink, ncu, iters);
    run<3, 2, 1, 1>("12 tr + 2 LDS-DMA, wait after MFMAs", src, out, sink, ncu, iters);
    run<3, 2, 0, 1>("12 tr + 2 LDS-DMA, wait after MFMAs, no setprio", src, out, sink, ncu, iters);
    return 0;
}
