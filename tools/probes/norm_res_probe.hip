// VERDICT r5 item 3, measured: the "x-hat" form of the post-LN sub-layer.  Today:  h = x + drop(sub(x)) comes out of the GEMM epilogue (gemm_nt8p<0,1,true>,
// residual + dropout), then norm_fwd reads h (1 KB / row) and writes LN(h) (1 KB).  x-hat form: the GEMM writes s = sub(x) + bias with the plain assembly kernels, and
// THIS kernel forms z = (xhat_prev * gamma_prev + beta_prev) + drop(s) itself -- reads s and xhat_prev (2 KB / row), writes xhat = (z - mu) * rstd (1 KB) and the row
// statistics; gamma / beta of THIS norm are folded into the consuming weights.  Built on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include -I safevla_amd/csrc tools/probes/norm_res_probe.hip -o /tmp/norm_res_probe.so
// (same two-rows-per-trip structure, non-temporal row loads / stores and dropout counter hash as csrc/norm.hip's norm_fwd_kernel; timing probe, not product code)
#include "common.h"
template <int D>
__global__ void norm_res_fwd_kernel(const bf16_t* __restrict__ s, const bf16_t* __restrict__ xhat_prev, const float* __restrict__ gamma_prev,
                                    const float* __restrict__ beta_prev, float eps, int rows, bf16_t* __restrict__ xhat, float* __restrict__ mean_out,
                                    float* __restrict__ rstd_out, DropCfg drop) {
    constexpr int VPL = D / 64;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    float g[VPL], b[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { g[i] = gamma_prev[lane * VPL + i]; b[i] = beta_prev[lane * VPL + i]; }
    auto ld = [&](const bf16_t* p, float (&v)[VPL]) {
        const u32x4 w = __builtin_nontemporal_load((const u32x4*)p);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf_lo(w[i]); v[2 * i + 1] = bf_hi(w[i]); }
    };
    for (int m = 2 * wave; m < rows; m += 2 * nw) {
        float sv[2][VPL], xv[2][VPL];
#pragma unroll
        for (int j = 0; j < 2; ++j) if (m + j < rows) { ld(s + (size_t)(m + j) * D + lane * VPL, sv[j]); ld(xhat_prev + (size_t)(m + j) * D + lane * VPL, xv[j]); }
#pragma unroll
        for (int j = 0; j < 2; ++j) if (m + j < rows) {
            float v[VPL];
            unsigned keep = 0xffu;
            if (drop.thr) {
                const unsigned long long e0 = (unsigned long long)(m + j) * D + lane * VPL;
                keep = drop_keep4(drop, e0) | (drop_keep4(drop, e0 + 4) << 4);
            }
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const float y = fmaf(xv[j][i], g[i], b[i]);
                v[i] = y + (((keep >> i) & 1u) ? sv[j][i] * drop.scale : 0.f);
                sum += v[i];
            }
            const float mu = wave_sum(sum) * (1.f / D);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { const float d = v[i] - mu; q += d * d; }
            const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
            if (lane == 0) { mean_out[m + j] = mu; rstd_out[m + j] = rstd; }
            u32x4 w;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = pack_bf2((v[2 * i] - mu) * rstd, (v[2 * i + 1] - mu) * rstd);
            __builtin_nontemporal_store(w, (u32x4*)(xhat + (size_t)(m + j) * D + lane * VPL));
        }
    }
}
extern "C" int norm_res_fwd(const void* s, const void* xhat_prev, const float* gamma_prev, const float* beta_prev, int rows, void* xhat, float* mean, float* rstd,
                            unsigned seed, float p, void* stream) {
    svla_dropout d{seed, 5u, p, 1, nullptr};
    long blocks = ((long)rows + 7) / 8;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((norm_res_fwd_kernel<512>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)s, (const bf16_t*)xhat_prev, gamma_prev, beta_prev,
                       1e-5f, rows, (bf16_t*)xhat, mean, rstd, drop_cfg(&d));
    return (int)hipGetLastError();
}
