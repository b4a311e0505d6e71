// LayerNorm / RMSNorm forward+backward, bf16 activations, fp32 statistics and parameters.  HBM-bound:
// one wave per row (D/64 contiguous values per lane, 16-byte loads for D=512), grid-stride over rows.
//
// Reference ops: nn.LayerNorm inside nn.TransformerEncoderLayer (post-LN, eps 1e-5), the adapter
// "Linear -> LayerNorm -> ReLU (+ camera token)" stacks (allenact_dino_transformer.py:509-513,539-543,672-688),
// llama RMSNorm (training/online/third_party_models/llama/model.py:28-71, eps 1e-5), T5 RMS norm (eps 1e-6).
//
// Row maps: logical row m lives at memory row (m / G) * GS + OFF + (m % G) (G = 0: identity).  This lets the
// adapter LayerNorm write straight into (and its backward read straight out of) the token slice of the
// fusion-transformer input [R, S, D] without a concat/split copy.
#include "common.h"
#include <cstdlib>
#ifndef NORM_NT
#define NORM_NT 1
#endif

struct RowMap { int G, GS, OFF; };
__device__ __forceinline__ size_t map_row(const RowMap& rm, int m) {
    return rm.G > 0 ? (size_t)(m / rm.G) * rm.GS + rm.OFF + (m % rm.G) : (size_t)m;
}

template <int VPL>
__device__ __forceinline__ void load_row_bf16(const bf16_t* p, float (&v)[VPL]) {
    if constexpr (VPL == 8) {
        const u32x4 w = NORM_NT ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf_lo(w[i]); v[2 * i + 1] = bf_hi(w[i]); }
    } else {
        const uint32_t* q = (const uint32_t*)p;
#pragma unroll
        for (int i = 0; i < VPL / 2; ++i) { const uint32_t w = q[i]; v[2 * i] = bf_lo(w); v[2 * i + 1] = bf_hi(w); }
    }
}
template <int VPL>
__device__ __forceinline__ void store_row_bf16(bf16_t* p, const float (&v)[VPL]) {
    if constexpr (VPL == 8) {
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
        if (NORM_NT) __builtin_nontemporal_store(w, (u32x4*)p); else *(u32x4*)p = w;
    } else {
        uint32_t* q = (uint32_t*)p;
#pragma unroll
        for (int i = 0; i < VPL / 2; ++i) q[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
    }
}

// fp32 activations (the fp32 verification mode): the same kernels instantiated on float rows
template <int VPL>
__device__ __forceinline__ void load_row(const bf16_t* p, float (&v)[VPL]) { load_row_bf16<VPL>(p, v); }
template <int VPL>
__device__ __forceinline__ void store_row(bf16_t* p, const float (&v)[VPL]) { store_row_bf16<VPL>(p, v); }
template <int VPL>
__device__ __forceinline__ void load_row(const float* p, float (&v)[VPL]) {
#pragma unroll
    for (int i = 0; i < VPL / 2; ++i) { const float2 w = *(const float2*)(p + 2 * i); v[2 * i] = w.x; v[2 * i + 1] = w.y; }
}
template <int VPL>
__device__ __forceinline__ void store_row(float* p, const float (&v)[VPL]) {
#pragma unroll
    for (int i = 0; i < VPL / 2; ++i) *(float2*)(p + 2 * i) = float2{v[2 * i], v[2 * i + 1]};
}

// ---------------------------------------------------------------------------------------------- LayerNorm fwd
// y = [relu](LN(x) * gamma + beta) [+ tok[(m % G) / tok_group]]        (rms != 0: RMS norm, beta ignored)
template <typename T, int D>
__device__ __forceinline__ void norm_fwd_kernel_body(const T* __restrict__ x, RowMap xmap, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int rows, int rms, int relu,
                                const float* __restrict__ tok, int tok_group, T* __restrict__ y, RowMap ymap,
                                float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    constexpr int VPL = D / 64;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    float g[VPL], b[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { g[i] = gamma[lane * VPL + i]; b[i] = (beta && !rms) ? beta[lane * VPL + i] : 0.f; }
    // two rows per trip, both loads issued before the first reduction: one 1-KiB load per wave in flight does not cover the HBM
    // latency-bandwidth product (3.9 TB/s with one row per trip, 4.6 with two, 3.9 again with four: fewer waves per SIMD)
    auto finish = [&](int m, float (&v)[VPL]) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) s += v[i];
        const float mu = rms ? 0.f : wave_sum(s) * (1.f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) { const float d = v[i] - mu; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) * (1.f / D) + eps);
        if (lane == 0) { if (mean_out) mean_out[m] = mu; if (rstd_out) rstd_out[m] = rstd; }
        const float* tk = tok ? tok + (size_t)((ymap.G > 0 ? (m % ymap.G) : m) / tok_group) * D + lane * VPL : nullptr;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float o = (v[i] - mu) * rstd * g[i] + b[i];
            if (relu) o = fmaxf(o, 0.f);
            if (tk) o += tk[i];
            v[i] = o;
        }
        store_row<VPL>(y + map_row(ymap, m) * D + lane * VPL, v);
    };
#ifndef NORM_RPT
#define NORM_RPT 2
#endif
    constexpr int RPT = NORM_RPT;
    for (int m = RPT * wave; m < rows; m += RPT * nw) {
        float v[RPT][VPL];
#pragma unroll
        for (int j = 0; j < RPT; ++j) if (m + j < rows) load_row<VPL>(x + map_row(xmap, m + j) * D + lane * VPL, v[j]);
#pragma unroll
        for (int j = 0; j < RPT; ++j) if (m + j < rows) finish(m + j, v[j]);
    }
}
template <typename T, int D>
__global__ void norm_fwd_kernel(const T* __restrict__ x, RowMap xmap, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float eps, int rows, int rms, int relu,
                                const float* __restrict__ tok, int tok_group, T* __restrict__ y, RowMap ymap,
                                float* __restrict__ mean_out, float* __restrict__ rstd_out) { norm_fwd_kernel_body<T, D>(x, xmap, gamma, beta, eps, rows, rms, relu, tok, tok_group, y, ymap, mean_out, rstd_out); }

// ---------------------------------------------------------------------------------------------- LayerNorm bwd
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma [* relu mask];   (rms: no mean(g) term)
// dgamma += sum dy_masked * xhat; dbeta += sum dy_masked; dtok[k] += sum dy (rows of token group k)
template <typename T, int D>
__global__ void norm_bwd_kernel(const T* __restrict__ dy, RowMap dymap, const T* __restrict__ x, RowMap xmap,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ mean, const float* __restrict__ rstd_in, int rows, int rms,
                                int relu, int tok_group, const T* __restrict__ dres, T* __restrict__ dx,
                                RowMap dxmap, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                float* __restrict__ dtok, T* __restrict__ dx_drop, DropCfg drop, DetCfg det) {
    drop = drop_resolve(drop);
    constexpr int VPL = D / 64;
    __shared__ float red[4][4 * D];  // [wave][dgamma | dbeta | dtok0 | dtok1]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    float g[VPL], b[VPL], ag[VPL], ab[VPL], at0[VPL], at1[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        g[i] = gamma[lane * VPL + i]; b[i] = (beta && !rms) ? beta[lane * VPL + i] : 0.f;
        ag[i] = ab[i] = at0[i] = at1[i] = 0.f;
    }
    auto finish = [&](int m, float (&xv)[VPL], float (&dv)[VPL], float mu, float rs) {
        if (dtok) {
            const int k = (dymap.G > 0 ? (m % dymap.G) : m) / tok_group;
#pragma unroll
            for (int i = 0; i < VPL; ++i) { if (k == 0) at0[i] += dv[i]; else at1[i] += dv[i]; }
        }
        float s1 = 0.f, s2 = 0.f, gg[VPL], xh[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            xh[i] = (xv[i] - mu) * rs;
            float d = dv[i];
            if (relu && !(xh[i] * g[i] + b[i] > 0.f)) d = 0.f;
            ag[i] += d * xh[i]; ab[i] += d;
            gg[i] = d * g[i];
            s1 += gg[i]; s2 += gg[i] * xh[i];
        }
        s1 = rms ? 0.f : wave_sum(s1) * (1.f / D);
        s2 = wave_sum(s2) * (1.f / D);
#pragma unroll
        for (int i = 0; i < VPL; ++i) xv[i] = rs * (gg[i] - s1 - xh[i] * s2);
        if (dres) {  // residual-branch gradient (pre-norm blocks: d h = d out + norm_bwd(...))
            float rv[VPL];
            load_row<VPL>(dres + (size_t)m * D + lane * VPL, rv);
#pragma unroll
            for (int i = 0; i < VPL; ++i) xv[i] += rv[i];
        }
        store_row<VPL>(dx + map_row(dxmap, m) * D + lane * VPL, xv);
        if (dx_drop) {
            // gradient through the dropout on the sub-layer output that was added to the residual stream before this norm
            // (x + dropout(sublayer(x))): the same keep-mask as the forward GEMM epilogue, regenerated from the element index
#pragma unroll
            for (int q4 = 0; q4 < VPL / 4; ++q4) {
                const unsigned keep = drop.thr ? drop_keep4(drop, (unsigned long long)m * drop.row_mult * D + lane * VPL + q4 * 4) : 0xfu;
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[q4 * 4 + e] = ((keep >> e) & 1u) ? xv[q4 * 4 + e] * drop.scale : 0.f;
            }
            store_row<VPL>(dx_drop + (size_t)m * D + lane * VPL, xv);
        }
    };
    // two rows per trip, all four row loads issued before the first reduction (see norm_fwd_kernel)
    for (int m = 2 * wave; m < rows; m += 2 * nw) {
        float x0[VPL], d0[VPL], x1[VPL], d1[VPL];
        const bool two = m + 1 < rows;
        load_row<VPL>(x + map_row(xmap, m) * D + lane * VPL, x0);
        load_row<VPL>(dy + map_row(dymap, m) * D + lane * VPL, d0);
        if (two) {
            load_row<VPL>(x + map_row(xmap, m + 1) * D + lane * VPL, x1);
            load_row<VPL>(dy + map_row(dymap, m + 1) * D + lane * VPL, d1);
        }
        const float mu0 = rms ? 0.f : mean[m], rs0 = rstd_in[m];
        const float mu1 = (rms || !two) ? 0.f : mean[m + 1], rs1 = two ? rstd_in[m + 1] : 0.f;
        finish(m, x0, d0, mu0, rs0);
        if (two) finish(m + 1, x1, d1, mu1, rs1);
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        red[wid][lane * VPL + i] = ag[i]; red[wid][D + lane * VPL + i] = ab[i];
        red[wid][2 * D + lane * VPL + i] = at0[i]; red[wid][3 * D + lane * VPL + i] = at1[i];
    }
    __syncthreads();
    const int nwv = blockDim.x >> 6;
    for (int c = threadIdx.x; c < 4 * D; c += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < nwv; ++w) s += red[w][c];
        if (c < D) { if (dgamma) grad_add(det, &dgamma[c], s); }
        else if (c < 2 * D) { if (dbeta && !rms) grad_add(det, &dbeta[c - D], s); }
        else if (dtok) grad_add(det, &dtok[c - 2 * D], s);
    }
}

static inline int norm_grid(int rows, int cap_default) {
    int blocks = (rows + 7) / 8;      // 4 waves x 2 rows per trip
    static int sweep = -1;
    if (sweep < 0) { const char* e = getenv("SVLA_NORM_GRID"); sweep = e ? atoi(e) : 0; }      // (tools/norm_bw.py: grid sweep)
    const int cap = sweep > 0 ? sweep : cap_default;
    return blocks > cap ? cap : (blocks < 1 ? 1 : blocks);
}

template <typename T>
static int norm_fwd_launch(const T* x, int xG, int xGS, int xOFF, const float* gamma, const float* beta, float eps, int rows, int D, int rms,
                           int relu, const float* tok, int tok_group, T* y, int yG, int yGS, int yOFF, float* mean, float* rstd, void* stream) {
    if (rows <= 0) return SVLA_EINVAL;
    if (tok && tok_group <= 0) return SVLA_EINVAL;
    RowMap xm{xG, xGS, xOFF}, ym{yG, yGS, yOFF};
    dim3 grid(norm_grid(rows, 2048)), block(256);
#define NORM_FWD_CASE(DD) SVLA_LAUNCH((norm_fwd_kernel<T, DD>), (norm_fwd_kernel_body<T, DD>), 1024, 1, grid, block, 0, (hipStream_t)stream, x, xm, gamma, beta, eps, rows, rms, relu, tok, tok_group, y, ym, mean, rstd)
    // widths on this path: 512 (policy, T5), 384 / 768 / 1024 (frozen ViT-S / ViT-B + SigLIP-B / ViT-L preprocessors)
    if (D == 512) NORM_FWD_CASE(512);
    else if (D == 384) NORM_FWD_CASE(384);
    else if (D == 768) NORM_FWD_CASE(768);
    else if (D == 1024) NORM_FWD_CASE(1024);
    else return SVLA_EINVAL;
#undef NORM_FWD_CASE
    return svla_launch_status();
}

template <typename T>
static int norm_bwd_launch(const T* dy, int dyG, int dyGS, int dyOFF, const T* x, int xG, int xGS, int xOFF, const float* gamma,
                           const float* beta, const float* mean, const float* rstd, int rows, int D, int rms, int relu, int tok_group,
                           const T* dres, T* dx, int dxG, int dxGS, int dxOFF, float* dgamma, float* dbeta, float* dtok, T* dx_drop,
                           const svla_dropout* drop, void* stream) {
    if (rows <= 0 || (D != 512 && D != 768)) return SVLA_EINVAL;
    if (dtok && tok_group <= 0) return SVLA_EINVAL;
    RowMap dym{dyG, dyGS, dyOFF}, xm{xG, xGS, xOFF}, dxm{dxG, dxGS, dxOFF};
    const int blocks = norm_grid(rows, 4096);   // measured (tools/norm_bw.py, 2.97 M rows): 5.09 TB/s at 1 024 workgroups, 5.23 at 4 096 (each ends with up to 4*D atomics)
#define NORM_BWD_CASE(DD) hipLaunchKernelGGL((norm_bwd_kernel<T, DD>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, dym, x, xm, gamma, beta, mean, \
                       rstd, rows, rms, relu, tok_group > 0 ? tok_group : 1, dres, dx, dxm, dgamma, dbeta, dtok, dx_drop, drop_cfg(drop), g_svla_det)
    if (D == 512) NORM_BWD_CASE(512);       // the policy
    else NORM_BWD_CASE(768);                // the 768-wide imitation-learning presets (early_fusion_tsfm_models.py:275-294)
#undef NORM_BWD_CASE
    return svla_launch_status();
}

extern "C" int svla_norm_fwd_bf16(const bf16_t* x, int xG, int xGS, int xOFF, const float* gamma, const float* beta, float eps,
                                  int rows, int D, int rms, int relu, const float* tok, int tok_group, bf16_t* y, int yG,
                                  int yGS, int yOFF, float* mean, float* rstd, void* stream) {
    return norm_fwd_launch<bf16_t>(x, xG, xGS, xOFF, gamma, beta, eps, rows, D, rms, relu, tok, tok_group, y, yG, yGS, yOFF, mean, rstd, stream);
}
extern "C" int svla_norm_fwd_f32(const float* x, int xG, int xGS, int xOFF, const float* gamma, const float* beta, float eps,
                                 int rows, int D, int rms, int relu, const float* tok, int tok_group, float* y, int yG,
                                 int yGS, int yOFF, float* mean, float* rstd, void* stream) {
    return norm_fwd_launch<float>(x, xG, xGS, xOFF, gamma, beta, eps, rows, D, rms, relu, tok, tok_group, y, yG, yGS, yOFF, mean, rstd, stream);
}
extern "C" int svla_norm_bwd_bf16(const bf16_t* dy, int dyG, int dyGS, int dyOFF, const bf16_t* x, int xG, int xGS, int xOFF,
                                  const float* gamma, const float* beta, const float* mean, const float* rstd, int rows,
                                  int D, int rms, int relu, int tok_group, const bf16_t* dres, bf16_t* dx, int dxG, int dxGS,
                                  int dxOFF, float* dgamma, float* dbeta, float* dtok, bf16_t* dx_drop, const svla_dropout* drop,
                                  void* stream) {
    return norm_bwd_launch<bf16_t>(dy, dyG, dyGS, dyOFF, x, xG, xGS, xOFF, gamma, beta, mean, rstd, rows, D, rms, relu, tok_group, dres, dx,
                                   dxG, dxGS, dxOFF, dgamma, dbeta, dtok, dx_drop, drop, stream);
}
extern "C" int svla_norm_bwd_f32(const float* dy, int dyG, int dyGS, int dyOFF, const float* x, int xG, int xGS, int xOFF,
                                 const float* gamma, const float* beta, const float* mean, const float* rstd, int rows,
                                 int D, int rms, int relu, int tok_group, const float* dres, float* dx, int dxG, int dxGS,
                                 int dxOFF, float* dgamma, float* dbeta, float* dtok, float* dx_drop, const svla_dropout* drop,
                                 void* stream) {
    return norm_bwd_launch<float>(dy, dyG, dyGS, dyOFF, x, xG, xGS, xOFF, gamma, beta, mean, rstd, rows, D, rms, relu, tok_group, dres, dx,
                                  dxG, dxGS, dxOFF, dgamma, dbeta, dtok, dx_drop, drop, stream);
}
