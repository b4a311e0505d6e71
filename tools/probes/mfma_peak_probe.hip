// What does the box sustain on MFMA alone?  Register-only v_mfma_f32_32x32x16_bf16 loop, 8 waves per CU (2 per SIMD), every CU busy,
// no memory traffic.  Prints wall-clock TFLOP/s (HIP events), shader cycles per MFMA per SIMD (32 = issue-bound) and the shader clock
// the two imply.  The 2.5 PFLOP/s dense-bf16 figure of the microarchitecture guide assumes 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak_probe mfma_peak_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_hw;

template <int NACC, int RANDOM>
__global__ void __launch_bounds__(512, 2) probe(int iters, long long* cyc, float* sink) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_hw a[4], b[2];
    uint32_t h = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            h = h * 1664525u + 1013904223u;
            // RANDOM: values with random sign and mantissa in [-2, 2) (what activations x weights look like to the multiplier array);
            // otherwise small constants (few toggling bits)
            const float v = RANDOM ? ((float)(int)(h >> 8) * (1.f / 8388608.f) - 1.f) * 2.f : (float)(threadIdx.x & 3);
            a[q][r] = (__bf16)v;
            if (q < 2) b[q][r] = RANDOM ? (__bf16)(((float)(int)((h * 7u) >> 8) * (1.f / 8388608.f) - 1.f) * 0.05f) : (__bf16)0.5f;
        }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 2) & 1], acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    const int secs_iters = argc > 1 ? atoi(argv[1]) : 200000;
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    long long* cyc; float* sink;
    hipMalloc(&cyc, sizeof(long long) * ncu * 2); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int random = 0; random <= 1; ++random)
        for (int rep = 0; rep < 4; ++rep) {
            const int wgs_per_cu = 1;
            const int iters = secs_iters;
            hipEventRecord(e0);
            if (random) probe<8, 1><<<ncu, 512>>>(iters, cyc, sink); else probe<8, 0><<<ncu, 512>>>(iters, cyc, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[512]; hipMemcpy(h, cyc, sizeof(long long) * ncu * wgs_per_cu, hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < ncu * wgs_per_cu; ++i) avg += (double)h[i]; avg /= ncu * wgs_per_cu;
            const double mfmas = (double)iters * 8;                       // per wave
            const double flops = mfmas * 32768.0 * 8 * ncu * wgs_per_cu;  // 8 waves per workgroup
            const double cyc_per_mfma_simd = avg / (mfmas * 2 * wgs_per_cu);
            printf("%s operands, %d CUs x %d workgroup(s) of 8 waves, %d x 8 MFMAs per wave: %.2f ms  %.0f TFLOP/s wall-clock; %.1f shader cycles per MFMA per SIMD; implied shader clock %.2f GHz\n",
                   random ? "random" : "constant", ncu, wgs_per_cu, iters, ms, flops / ms * 1e-9, cyc_per_mfma_simd, avg / ms * 1e-6);
        }
    return 0;
}
