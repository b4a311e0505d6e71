// "where did this workgroup run" probe for tools/cu_split_probe.py (built on the GPU box: hipcc --offload-arch=gfx950 -O2 -shared -fPIC whereami_probe.hip -o /tmp/whereami.so):
// every workgroup records the XCC it ran on and its HW_ID word (CU / SH / SE fields), then spins ~spin_cycles so that a large grid spreads over every CU the
// stream's CU mask allows.  Used to find out how the bits of hipExtStreamCreateWithCUMask map to XCDs on the MI355X.
#include <hip/hip_runtime.h>
extern "C" __global__ void whereami_kernel(unsigned* out, long long spin_cycles) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin_cycles) { __builtin_amdgcn_s_sleep(8); }
}
extern "C" int whereami_launch(unsigned* out, int grid, long long spin_cycles, void* stream) {
    hipLaunchKernelGGL(whereami_kernel, dim3(grid), dim3(64), 0, (hipStream_t)stream, out, spin_cycles);
    return (int)hipGetLastError();
}
extern "C" int make_masked_stream(void** stream, unsigned n_words, const unsigned* mask) {
    hipStream_t s;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, n_words, mask);
    *stream = (void*)s;
    return (int)e;
}
