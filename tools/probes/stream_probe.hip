// HBM streaming calibration: hand-written copy (1 read + 1 write), 2-reads-1-write and read-only kernels, 16 B per lane,
// persistent grid-stride loops with U independent 16-byte loads in flight per lane, plain and non-temporal; 2 GiB per stream so nothing is cache
// resident.  Prints TB/s of (bytes read + bytes written).  Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int U, bool NT>      // MODE 0: copy, 1: a + b -> c, 2: read-only (sum)
__global__ void __launch_bounds__(256) stream(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ c, long n, unsigned* sink) {
    const long stride = (long)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
        u32x4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride;
            if (j < n) {
                x[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
                if (MODE == 1) y[u] = NT ? __builtin_nontemporal_load(b + j) : b[j];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride;
            if (j < n) {
                if (MODE == 2) acc += x[u];
                else {
                    const u32x4 v = MODE == 1 ? x[u] + y[u] : x[u];
                    if (NT) __builtin_nontemporal_store(v, c + j); else c[j] = v;
                }
            }
        }
    }
    if (MODE == 2 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) *sink = 1;
}

template <int MODE, int U, bool NT>
static void run(const char* name, const u32x4* a, const u32x4* b, u32x4* c, long n, unsigned* sink, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream<MODE, U, NT>), dim3(wgs), dim3(256), 0, 0, a, b, c, n, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)n * 16 * (MODE == 0 ? 2 : MODE == 1 ? 3 : 1);
    printf("%-28s U=%d %s grid=%5d: %7.3f ms  %5.2f TB/s\n", name, U, NT ? "nt   " : "plain", wgs, best, bytes / best * 1e-9);
}

int main() {
    const long n = (2L << 30) / 16;      // 2 GiB per stream
    u32x4 *a, *b, *c; unsigned* sink;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&sink, 4);
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16); hipMemset(c, 0, n * 16);
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    for (int per_cu : {4, 8, 16}) {
        const int wgs = ncu * per_cu;
        run<0, 1, false>("copy (1R + 1W)", a, b, c, n, sink, wgs);
        run<0, 4, false>("copy (1R + 1W)", a, b, c, n, sink, wgs);
        run<0, 4, true>("copy (1R + 1W)", a, b, c, n, sink, wgs);
        run<0, 8, true>("copy (1R + 1W)", a, b, c, n, sink, wgs);
        run<1, 2, false>("2 reads + 1 write", a, b, c, n, sink, wgs);
        run<1, 4, true>("2 reads + 1 write", a, b, c, n, sink, wgs);
        run<2, 4, false>("read only", a, b, c, n, sink, wgs);
        run<2, 8, true>("read only", a, b, c, n, sink, wgs);
    }
    return 0;
}
