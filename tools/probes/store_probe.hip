// Per-CU global-store throughput by store shape (one 512-thread workgroup per CU, 8 waves, each wave writes its own region of a big
// buffer once, streaming: nothing is re-written, so the rates are what an epilogue sees).  Shapes per wave instruction (64 lanes x 16 B = 1 KiB):
//   8 rows x 128 B (the GEMM epilogue's), 4 x 256 B, 2 x 512 B, 1 x 1024 B contiguous; plus dwordx2 (8 B/lane) for reference.  Row pitch 4 KiB.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

template <int ROWS, int W>      // ROWS rows per instruction, W = bytes per lane (16 or 8)
__global__ void __launch_bounds__(512, 2) probe(char* __restrict__ dst, int iters, long long* out) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lanes_per_row = 64 / ROWS, seg = lanes_per_row * W;      // bytes per row segment
    const size_t pitch = 4096;
    char* base = dst + ((size_t)blockIdx.x * 8 + wid) * ((size_t)iters * 16 * pitch * (ROWS > 8 ? ROWS : 8));
    u32x4 v = {(uint32_t)tid, 1u, 2u, 3u};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            char* p = base + ((size_t)(it * 16 + k) * ROWS + lane / lanes_per_row) * pitch + (size_t)(lane % lanes_per_row) * W;
            if (W == 16) *(u32x4*)p = v; else *(u32x2*)p = u32x2{v[0], v[1]};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) out[blockIdx.x] = t1 - t0;
    (void)seg;
}

template <int ROWS, int W>
static void run(const char* name, char* dst, long long* out, int ncu, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<ROWS, W><<<ncu, 512>>>(dst, 2, out); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); probe<ROWS, W><<<ncu, 512>>>(dst, iters, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
    const double bytes_cu = 8.0 * iters * 16 * 64 * W;
    printf("%-44s %8.3f ms  %6.2f TB/s chip  %6.1f B/clk/CU (shader cycles)  %5.1f cycles per wave-instruction per CU\n", name, ms,
           bytes_cu * ncu / ms / 1e9, bytes_cu / (double)c, (double)c / (8.0 * iters * 16));
}

int main() {
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount, iters = 64;
    char* dst; long long* out;
    const size_t bytes = (size_t)ncu * 8 * iters * 16 * 4096 * 8 + (1 << 20);
    (void)hipMalloc(&dst, bytes); (void)hipMalloc(&out, ncu * 8);
    printf("buffer %.1f GB, %d CUs\n", bytes / 1e9, ncu);
    run<8, 16>("dwordx4, 8 rows x 128 B per instruction", dst, out, ncu, iters);
    run<4, 16>("dwordx4, 4 rows x 256 B", dst, out, ncu, iters);
    run<2, 16>("dwordx4, 2 rows x 512 B", dst, out, ncu, iters);
    run<1, 16>("dwordx4, 1 KiB contiguous", dst, out, ncu, iters);
    run<8, 8>("dwordx2, 8 rows x 64 B", dst, out, ncu, iters);
    run<1, 8>("dwordx2, 512 B contiguous", dst, out, ncu, iters);
    return 0;
}
