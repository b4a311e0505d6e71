// s_barrier cost on gfx950 for a 512-thread (8-wave) workgroup, one per CU: (1) bare barriers, (2) barrier + 8 MFMAs per interval with
// the two wave groups in lockstep, (3) the ping-pong pattern (groups one barrier apart, 8 MFMAs every other interval), (4) ping-pong
// with 16 MFMAs per compute interval.  Prints cycles per barrier interval (s_memtime) and implied MFMA utilisation.
// Build: hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_hw;

template <int MODE, int NM>
__global__ void __launch_bounds__(512, 2) probe(int iters, long long* out, float* sink) {
    const int wid = threadIdx.x >> 6;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)threadIdx.x;
    bf16x8_hw a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
    if (MODE >= 3 && wid >= 4) __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            asm volatile("s_barrier" ::: "memory");
        } else if (MODE == 2) {
            asm volatile("s_barrier" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])::"memory");
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 7], 0, 0, 0);
        } else {
            asm volatile("s_barrier\n\ts_setprio 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])::"memory");
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 7], 0, 0, 0);
            asm volatile("s_setprio 0\n\ts_barrier" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])::"memory");
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (MODE >= 3 && wid < 4) __builtin_amdgcn_s_barrier();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE, int NM>
static void run(const char* name, long long* out, float* sink, int ncu, int iters, int barriers_per_iter, int mfma_per_simd_per_iter) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE, NM><<<ncu, 512>>>(10, out, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<MODE, NM><<<ncu, 512>>>(iters, out, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    const double ns_per_iter = ms * 1e6 / iters;
    printf("%-52s %7.1f ns/iter  %6.1f ns per barrier interval  s_memtime ticks/iter %6.1f   MFMA pipe busy (32 clk each @2.4GHz) %4.1f %%\n", name, ns_per_iter,
           ns_per_iter / barriers_per_iter, (double)c / iters, 100.0 * mfma_per_simd_per_iter * 32 / 2.4 / ns_per_iter);
}

int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    long long* out; float* sink;
    (void)hipMalloc(&out, ncu * 8); (void)hipMalloc(&sink, 4096);
    const int iters = 20000;
    run<1, 0>("bare s_barrier (8 waves)", out, sink, ncu, iters, 1, 0);
    run<2, 8>("barrier + 8 MFMA, all waves lockstep", out, sink, ncu, iters, 1, 16);
    run<2, 16>("barrier + 16 MFMA, lockstep", out, sink, ncu, iters, 1, 32);
    run<2, 32>("barrier + 32 MFMA, lockstep", out, sink, ncu, iters, 1, 64);
    run<3, 8>("ping-pong: 2 barriers/iter, 8 MFMA per group", out, sink, ncu, iters, 2, 16);
    run<3, 16>("ping-pong: 2 barriers/iter, 16 MFMA per group", out, sink, ncu, iters, 2, 32);
    run<3, 32>("ping-pong: 2 barriers/iter, 32 MFMA per group", out, sink, ncu, iters, 2, 64);
    return 0;
}
