// HBM-bound glue kernels of the policy forward/backward and the optimiser step (gfx950).
//   feat_to_tokens      : (R,384,7,12) f32 channels-first DINO features -> bf16 token-major [R,2,84,384]
//   fusion_fill / _bwd  : fusion token + per-episode text tokens into / out of the fusion input [R,S,512]
//   decoder_embed / _bwd: beliefs input = fusion[:,0] + prev-action emb + in-hand emb + sinusoidal time enc
//   swiglu / _bwd       : llama FeedForward gate
//   adam / sumsq / cast / transpose : flat-buffer optimiser step (clip by global norm, no host sync)
#include "common.h"
#include <mutex>
#include <type_traits>

// ------------------------------------------------------------------------------------------------
// DataAugmentation/DINO preprocessor output is (R, C=384, 7, 12) fp32 (dino_preprocessors.py:31-35); the 1x1-conv
// compressor wants tokens with the channel (reduction) index contiguous.  Transpose through LDS, cast to bf16.
__global__ void feat_to_tokens_kernel(const float* __restrict__ feat, int R, int C, int P, int cam, int ncam,
                                      bf16_t* __restrict__ out) {
    __shared__ float tile[64][85];
    const int r = blockIdx.x, c0 = blockIdx.y * 64;
    const float* src = feat + ((size_t)r * C + c0) * P;
    for (int i = threadIdx.x; i < 64 * P; i += blockDim.x) tile[i / P][i % P] = src[i];
    __syncthreads();
    bf16_t* dst = out + ((size_t)(r * ncam + cam) * P) * C + c0;
    for (int i = threadIdx.x; i < P * 32; i += blockDim.x) {
        const int pp = i >> 5, cp = i & 31;
        *(uint32_t*)(dst + (size_t)pp * C + 2 * cp) = pack_bf2(tile[2 * cp][pp], tile[2 * cp + 1][pp]);
    }
}

extern "C" int svla_feat_to_tokens(const float* feat, int R, int C, int P, int cam, int ncam, bf16_t* out, void* stream) {
    if (R <= 0 || (C % 64) || P > 85 || cam >= ncam) return SVLA_EINVAL;
    hipLaunchKernelGGL(feat_to_tokens_kernel, dim3(R, C / 64), dim3(256), 0, (hipStream_t)stream, feat, R, C, P, cam, ncam, out);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// x0[r, 0, :] = fusion_token; x0[r, text_off + j, :] = text[gid[r], j, :]   (allenact_dino_transformer.py:672-692)
__device__ __forceinline__ void fusion_fill_kernel_body(const float* __restrict__ fusion_token, const bf16_t* __restrict__ text,
                                   const int* __restrict__ gid, int R, int S, int L, int text_off, int D, bf16_t* __restrict__ x0) {
    const int r = blockIdx.x, lane = threadIdx.x;  // 64 threads x 8 elements per 512-wide slice of the row (D = 512: one slice)
    bf16_t* row = x0 + (size_t)r * S * D;
    const bf16_t* tsrc = text + (size_t)gid[r] * L * D;
    for (int c = lane * 8; c < D; c += 512) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf2(fusion_token[c + 2 * e], fusion_token[c + 2 * e + 1]);
        *(u32x4*)(row + c) = w;
        for (int j = 0; j < L; ++j)
            *(u32x4*)(row + (size_t)(text_off + j) * D + c) = *(const u32x4*)(tsrc + (size_t)j * D + c);
    }
}
__global__ void fusion_fill_kernel(const float* __restrict__ fusion_token, const bf16_t* __restrict__ text,
                                   const int* __restrict__ gid, int R, int S, int L, int text_off, int D, bf16_t* __restrict__ x0) { fusion_fill_kernel_body(fusion_token, text, gid, R, S, L, text_off, D, x0); }

extern "C" int svla_fusion_fill(const float* fusion_token, const bf16_t* text, const int* gid, int R, int S, int L,
                                int text_off, int D, bf16_t* x0, void* stream) {
    if (R <= 0 || text_off + L > S || D <= 0 || (D % 8)) return SVLA_EINVAL;
    SVLA_LAUNCH(fusion_fill_kernel, fusion_fill_kernel_body, 1024, 1, dim3(R), dim3(64), 0, (hipStream_t)stream, fusion_token, text, gid, R, S, L, text_off, D, x0);
    return svla_launch_status();
}

// dtext[gid[r], j, :] += dx0[r, text_off + j, :].  Rows are (t*B + b); an env's goal is constant over an episode,
// so one workgroup walks one env over t and flushes a register accumulator only when the goal id changes.
__global__ void fusion_text_bwd_kernel(const bf16_t* __restrict__ dx0, const int* __restrict__ gid, int T, int B, int S,
                                       int L, int text_off, int D, float* __restrict__ dtext, DetCfg det) {
    const int b = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    const int c = blockIdx.z * 512 + lane * 8;      // one 512-wide slice of the row per blockIdx.z (D = 512: one slice)
    if (c >= D) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cur = gid[b];
    for (int t = 0; t < T; ++t) {
        const int r = t * B + b;
        const int g = gid[r];
        if (g != cur) {
            float* d = dtext + ((size_t)cur * L + j) * D + c;
#pragma unroll
            for (int e = 0; e < 8; ++e) { grad_add(det, d + e, acc[e]); acc[e] = 0.f; }
            cur = g;
        }
        const u32x4 w = *(const u32x4*)(dx0 + ((size_t)r * S + text_off + j) * D + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += bf_lo(w[e]); acc[2 * e + 1] += bf_hi(w[e]); }
    }
    float* d = dtext + ((size_t)cur * L + j) * D + c;
#pragma unroll
    for (int e = 0; e < 8; ++e) grad_add(det, d + e, acc[e]);
}

extern "C" int svla_fusion_text_bwd(const bf16_t* dx0, const int* gid, int T, int B, int S, int L, int text_off, int D, float* dtext,
                                    void* stream) {
    if (T <= 0 || B <= 0 || L <= 0 || D <= 0 || (D % 8)) return SVLA_EINVAL;
    hipLaunchKernelGGL(fusion_text_bwd_kernel, dim3(B, L, (D + 511) / 512), dim3(64), 0, (hipStream_t)stream, dx0, gid, T, B, S, L, text_off, D, dtext, g_svla_det);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Decoder input (allenact_dino_transformer.py:353-385 + text_cond_visual_encoder.py:263-283):
//   out[b*T + t, :] = xf[(t*B+b)*S*512 + :]  (fusion token output)  + act_tab[masks ? prev_action : A] + hand_tab[hand]
//                   + pe(time_step),  pe[2i] = sin(pos*div[i]), pe[2i+1] = cos(pos*div[i])
__device__ __forceinline__ void decoder_embed_kernel_body(const bf16_t* __restrict__ xf, long xf_row_stride, const float* __restrict__ act_tab,
                                     const float* __restrict__ hand_tab, const float* __restrict__ div_term,
                                     const int64_t* __restrict__ prev_actions, const float* __restrict__ masks,
                                     const int64_t* __restrict__ hand, const int64_t* __restrict__ time_step, int T, int B,
                                     int n_actions, int D, bf16_t* __restrict__ out) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= T * B) return;
    const int t = wave / B, b = wave % B;
    const int64_t a = masks[wave] != 0.f ? prev_actions[wave] : (int64_t)n_actions;
    const float pos = (float)time_step[wave];
    for (int c = lane * 8; c < D; c += 512) {       // D = 512: one trip
        const u32x4 w = *(const u32x4*)(xf + (size_t)wave * xf_row_stride + c);
        const float* at = act_tab + (size_t)a * D + c;
        const float* ht = hand_tab + (size_t)hand[wave] * D + c;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ang = pos * div_term[c / 2 + e];
            // reference order: time_enc + ((obs + prev_action_emb) + in_hand_emb)
            v[2 * e] = sinf(ang) + ((bf_lo(w[e]) + at[2 * e]) + ht[2 * e]);
            v[2 * e + 1] = cosf(ang) + ((bf_hi(w[e]) + at[2 * e + 1]) + ht[2 * e + 1]);
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        *(u32x4*)(out + ((size_t)b * T + t) * D + c) = o;
    }
}
__global__ void decoder_embed_kernel(const bf16_t* __restrict__ xf, long xf_row_stride, const float* __restrict__ act_tab,
                                     const float* __restrict__ hand_tab, const float* __restrict__ div_term,
                                     const int64_t* __restrict__ prev_actions, const float* __restrict__ masks,
                                     const int64_t* __restrict__ hand, const int64_t* __restrict__ time_step, int T, int B,
                                     int n_actions, int D, bf16_t* __restrict__ out) { decoder_embed_kernel_body(xf, xf_row_stride, act_tab, hand_tab, div_term, prev_actions, masks, hand, time_step, T, B, n_actions, D, out); }

extern "C" int svla_decoder_embed_fwd(const bf16_t* xf, long xf_row_stride, const float* act_tab, const float* hand_tab,
                                      const float* div_term, const int64_t* prev_actions, const float* masks,
                                      const int64_t* hand, const int64_t* time_step, int T, int B, int n_actions, int D, bf16_t* out,
                                      void* stream) {
    if (T <= 0 || B <= 0 || D <= 0 || (D % 8)) return SVLA_EINVAL;
    SVLA_LAUNCH(decoder_embed_kernel, decoder_embed_kernel_body, 1024, 1, dim3((T * B + 3) / 4), dim3(256), 0, (hipStream_t)stream, xf, xf_row_stride, act_tab,
                       hand_tab, div_term, prev_actions, masks, hand, time_step, T, B, n_actions, D, out);
    return svla_launch_status();
}

// dxf[(t*B+b) row, :] = dout[b*T+t, :];  d act_tab / d hand_tab accumulated in LDS per block, then flushed.
// DET: the per-block LDS table is fixed-point too (the four waves of a block reach it in arbitrary order).
template <bool DET>
__global__ void decoder_embed_bwd_kernel(const bf16_t* __restrict__ dout, const int64_t* __restrict__ prev_actions,
                                         const float* __restrict__ masks, const int64_t* __restrict__ hand, int T, int B,
                                         int n_actions, int D, bf16_t* __restrict__ dxf, long dxf_row_stride,
                                         float* __restrict__ d_act_tab, float* __restrict__ d_hand_tab, DetCfg det) {
    extern __shared__ __attribute__((aligned(16))) char tab_raw[];  // [(n_actions + 2) + 3][D] float (or 64-bit fixed point)
    typedef typename std::conditional<DET, unsigned long long, float>::type acc_t;
    acc_t* tab = (acc_t*)tab_raw;
    const int nrows_tab = n_actions + 2 + 3;
    for (int i = threadIdx.x; i < nrows_tab * D; i += blockDim.x) tab[i] = (acc_t)0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    auto add = [&](int idx, float v) {
        if constexpr (DET) {
            if (fabsf(v) < det.max_partial) atomicAdd(&tab[idx], det_fixed(det, v));      // < 8192 rows per block: the table cannot wrap
            else {      // NaN / Inf / huge: straight to fp32, visible -- and counted (svla_det_bypass_count)
                atomicAdd(idx < (n_actions + 2) * D ? &d_act_tab[idx] : &d_hand_tab[idx - (n_actions + 2) * D], v);
                if (det.bypass) atomicAdd(det.bypass, 1ull);
            }
        } else atomicAdd(&tab[idx], v);
    };
    for (int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave < T * B; wave += nw) {
        const int t = wave / B, b = wave % B;
        const int a = masks[wave] != 0.f ? (int)prev_actions[wave] : n_actions;
        const int hh = n_actions + 2 + (int)hand[wave];
        for (int c = lane * 8; c < D; c += 512) {       // D = 512: one trip
            const u32x4 w = *(const u32x4*)(dout + ((size_t)b * T + t) * D + c);
            *(u32x4*)(dxf + (size_t)wave * dxf_row_stride + c) = w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = bf_lo(w[e]), hi = bf_hi(w[e]);
                add(a * D + c + 2 * e, lo);
                add(a * D + c + 2 * e + 1, hi);
                add(hh * D + c + 2 * e, lo);
                add(hh * D + c + 2 * e + 1, hi);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nrows_tab * D; i += blockDim.x) {
        float v;
        if constexpr (DET) v = (float)((double)(long long)tab[i] * det.unscale); else v = tab[i];
        if (v != 0.f) {
            if (i < (n_actions + 2) * D) grad_add(det, &d_act_tab[i], v);
            else grad_add(det, &d_hand_tab[i - (n_actions + 2) * D], v);
        }
    }
}

extern "C" int svla_decoder_embed_bwd(const bf16_t* dout, const int64_t* prev_actions, const float* masks, const int64_t* hand,
                                      int T, int B, int n_actions, int D, bf16_t* dxf, long dxf_row_stride, float* d_act_tab,
                                      float* d_hand_tab, void* stream) {
    if (T <= 0 || B <= 0 || D <= 0 || (D % 8)) return SVLA_EINVAL;
    const bool det = g_svla_det.i64[0] != nullptr || g_svla_det.i64[1] != nullptr;
    const size_t lds = (size_t)(n_actions + 5) * D * (det ? sizeof(unsigned long long) : sizeof(float));
    if (lds > 160 * 1024) return SVLA_EINVAL;
    int blocks = (T * B + 63) / 64;
    if (blocks > 128) blocks = 128;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)decoder_embed_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)decoder_embed_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    if (det)
        hipLaunchKernelGGL(decoder_embed_bwd_kernel<true>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, dout, prev_actions, masks, hand,
                           T, B, n_actions, D, dxf, dxf_row_stride, d_act_tab, d_hand_tab, g_svla_det);
    else
        hipLaunchKernelGGL(decoder_embed_bwd_kernel<false>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, dout, prev_actions, masks, hand,
                           T, B, n_actions, D, dxf, dxf_row_stride, d_act_tab, d_hand_tab, g_svla_det);
    return svla_launch_status();
}

// ---- deterministic accumulation: configuration and fold-back (common.h: DetCfg) -------------------------------------------------
DetCfg g_svla_det = {{nullptr, nullptr}, {nullptr, nullptr}, {0, 0}, nullptr, 4503599627370496.f, 0.25f, 2.220446049250313e-16};      // grid 2^-52, partials < 2^-2
// frac_bits in [36, 52]: the shadow's grid becomes 2^-frac_bits and partials up to 2^(50 - frac_bits) enter it (8192 of them cannot wrap the int64).  Call between updates:
// shadows must be empty (folded back) when the grid changes.
extern "C" int svla_det_set_grid(int frac_bits) {
    if (frac_bits < 36 || frac_bits > 52) return SVLA_EINVAL;
    g_svla_det.scale = ldexpf(1.f, frac_bits);
    g_svla_det.unscale = ldexp(1.0, -frac_bits);
    g_svla_det.max_partial = ldexpf(1.f, 50 - frac_bits);
    return SVLA_OK;
}

extern "C" int svla_det_config(int slot, float* f32_base, long long* i64_shadow, long n) {
    if (slot < 0 || slot > 1 || n < 0 || ((f32_base == nullptr) != (i64_shadow == nullptr))) return SVLA_EINVAL;
    if (f32_base && !g_svla_det.bypass) {      // the bypass counter lives as long as the library
        unsigned long long* c = nullptr;
        HIP_CHECK_RET(hipMalloc(&c, sizeof(unsigned long long)));
        HIP_CHECK_RET(hipMemset(c, 0, sizeof(unsigned long long)));
        g_svla_det.bypass = c;
    }
    g_svla_det.f32[slot] = f32_base; g_svla_det.i64[slot] = i64_shadow; g_svla_det.n[slot] = f32_base ? n : 0;
    return SVLA_OK;
}
// partials that had a registered shadow but took the plain fp32 atomic (|partial| >= 0.25 or non-finite) since the last reset: a deterministic-mode run is bitwise
// repeatable iff this stays 0 (synchronises the device)
extern "C" int svla_det_bypass_count(unsigned long long* count, int reset) {
    if (!count) return SVLA_EINVAL;
    *count = 0;
    if (!g_svla_det.bypass) return SVLA_OK;
    HIP_CHECK_RET(hipDeviceSynchronize());
    HIP_CHECK_RET(hipMemcpy(count, g_svla_det.bypass, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIP_CHECK_RET(hipMemset(g_svla_det.bypass, 0, sizeof(unsigned long long)));
    return SVLA_OK;
}
__global__ void det_finalize_kernel(float* __restrict__ f, long long* __restrict__ s, long n, double unscale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long long v = s[i];
        if (v != 0) { f[i] += (float)((double)v * unscale); s[i] = 0; }
    }
}
extern "C" int svla_det_finalize(float* f32, long long* i64_shadow, long n, void* stream) {
    if (n <= 0 || !f32 || !i64_shadow) return SVLA_EINVAL;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(det_finalize_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, f32, i64_shadow, n, g_svla_det.unscale);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// llama FeedForward gate (llama/model.py:359-360): g = silu(a) * b with [a | b] = x.[w1 | w3]^T  (row = [a(Hd) | b(Hd)])
__device__ __forceinline__ void swiglu_fwd_kernel_body(const bf16_t* __restrict__ ab, long M, int Hd, bf16_t* __restrict__ g) {
    const long n = M * (Hd / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / (Hd / 8);
        const int c = (int)(i % (Hd / 8)) * 8;
        const u32x4 a = *(const u32x4*)(ab + m * 2 * Hd + c), b = *(const u32x4*)(ab + m * 2 * Hd + Hd + c);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = bf_lo(a[e]), a1 = bf_hi(a[e]);
            o[e] = pack_bf2(a0 / (1.f + __expf(-a0)) * bf_lo(b[e]), a1 / (1.f + __expf(-a1)) * bf_hi(b[e]));
        }
        *(u32x4*)(g + m * Hd + c) = o;
    }
}
__global__ void swiglu_fwd_kernel(const bf16_t* __restrict__ ab, long M, int Hd, bf16_t* __restrict__ g) { swiglu_fwd_kernel_body(ab, M, Hd, g); }
__global__ void swiglu_bwd_kernel(const bf16_t* __restrict__ ab, const bf16_t* __restrict__ dg, long M, int Hd,
                                  bf16_t* __restrict__ dab) {
    const long n = M * (Hd / 8);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / (Hd / 8);
        const int c = (int)(i % (Hd / 8)) * 8;
        const u32x4 a = *(const u32x4*)(ab + m * 2 * Hd + c), b = *(const u32x4*)(ab + m * 2 * Hd + Hd + c);
        const u32x4 d = *(const u32x4*)(dg + m * Hd + c);
        u32x4 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float av[2] = {bf_lo(a[e]), bf_hi(a[e])}, bv[2] = {bf_lo(b[e]), bf_hi(b[e])}, dv[2] = {bf_lo(d[e]), bf_hi(d[e])};
            float da[2], db[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float sg = 1.f / (1.f + __expf(-av[k]));
                da[k] = dv[k] * bv[k] * sg * (1.f + av[k] * (1.f - sg));
                db[k] = dv[k] * av[k] * sg;
            }
            oa[e] = pack_bf2(da[0], da[1]); ob[e] = pack_bf2(db[0], db[1]);
        }
        *(u32x4*)(dab + m * 2 * Hd + c) = oa;
        *(u32x4*)(dab + m * 2 * Hd + Hd + c) = ob;
    }
}
extern "C" int svla_swiglu_fwd(const bf16_t* ab, long M, int Hd, bf16_t* g, void* stream) {
    if (M <= 0 || (Hd % 8)) return SVLA_EINVAL;
    long blocks = (M * (Hd / 8) + 255) / 256; if (blocks > 2048) blocks = 2048;
    SVLA_LAUNCH(swiglu_fwd_kernel, swiglu_fwd_kernel_body, 1024, 1, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, ab, M, Hd, g);
    return svla_launch_status();
}
extern "C" int svla_swiglu_bwd(const bf16_t* ab, const bf16_t* dg, long M, int Hd, bf16_t* dab, void* stream) {
    if (M <= 0 || (Hd % 8)) return SVLA_EINVAL;
    long blocks = (M * (Hd / 8) + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, ab, dg, M, Hd, dab);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------ optimiser
// Order-independent (round 5): every block writes its partial into a scratch slot, the block that finishes LAST adds the partials in slot order and
// accumulates the total into *out -- the clip coefficient (hence every parameter after the Adam step) no longer depends on the arrival order of 1024 fp64
// atomics ("deterministic mode repeats gradients only").  Launches on one stream are ordered, so the per-tower calls accumulate in a fixed order too.
__global__ void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ out, double* __restrict__ partial, unsigned* __restrict__ done) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
    s = wave_sum(s);
    __shared__ float part[4];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partial[blockIdx.x], (double)part[0] + (double)part[1] + (double)part[2] + (double)part[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    double t = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) t += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ double red[256];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) { *out += red[0]; *done = 0u; }
}
// Scratch ([1024] partials + the arrival counter) is per (device, stream): two launches in flight on different streams -- the cost tower's extras next to an
// optimiser step, two engines in one process -- would otherwise share the partials and the counter and both return wrong norms.  Launches on ONE stream
// are ordered and reuse their slot; the counter is re-zeroed by a memset node in front of every launch, so a launch that died mid-way cannot poison the next one.
struct SumsqSlot { int dev; hipStream_t stream; double* scratch; };
static std::mutex g_sumsq_mu;
static SumsqSlot g_sumsq_slots[64];
static int g_sumsq_n = 0;
extern "C" int svla_sumsq_f32(const float* g, long n, double* out, void* stream) {
    if (n <= 0) return SVLA_EINVAL;
    long blocks = (n + 1023) / 1024; if (blocks > 1024) blocks = 1024;
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    double* scratch = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_sumsq_mu);
        for (int i = 0; i < g_sumsq_n; ++i)
            if (g_sumsq_slots[i].dev == dev && g_sumsq_slots[i].stream == (hipStream_t)stream) { scratch = g_sumsq_slots[i].scratch; break; }
        if (!scratch) {
            if (g_sumsq_n >= 64) return SVLA_EINVAL;                                  // (64 distinct (device, stream) pairs: far beyond three tower streams x 8 devices' worth of use in one process)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return SVLA_EINVAL;   // no allocation inside a capture: call once before
            HIP_CHECK_RET(hipMalloc(&scratch, 1025 * sizeof(double)));
            g_sumsq_slots[g_sumsq_n++] = SumsqSlot{dev, (hipStream_t)stream, scratch};
        }
    }
    HIP_CHECK_RET(hipMemsetAsync(scratch + 1024, 0, sizeof(double), (hipStream_t)stream));
    hipLaunchKernelGGL(sumsq_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, g, n, out, scratch, (unsigned*)(scratch + 1024));
    return svla_launch_status();
}

// torch.optim.Adam (amsgrad=False) / AdamW (decoupled ``weight_decay``: p *= 1 - lr*wd first) fused with clip_grad_norm_: the clip coefficient is read
// from the device-side squared norm (no host sync).  Also refreshes the bf16 mirror of the parameters.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            bf16_t* __restrict__ p_bf16, long n, float lr, float beta1, float beta2, float eps, float bc1,
                            float bc2_sqrt, const double* __restrict__ gnorm_sq, float max_norm, float grad_scale, float decay) {
    float clip = grad_scale;
    if (gnorm_sq && max_norm > 0.f) {
        const float tn = sqrtf((float)(*gnorm_sq)) * grad_scale;
        const float c = max_norm / (tn + 1e-6f);
        clip *= fminf(c, 1.f);
    }
    const float step = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = m[i] + (gi - m[i]) * (1.f - beta1);        // lerp_
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;    // mul_ + addcmul_
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        const float pi = p[i] * decay - step * (mi / denom);
        m[i] = mi; v[i] = vi; p[i] = pi;
        if (p_bf16) p_bf16[i] = f2bf(pi);
    }
}
extern "C" int svla_adam_step_f32(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, long n, float lr, float beta1,
                                  float beta2, float eps, int step, const double* gnorm_sq, float max_norm, float grad_scale,
                                  float weight_decay, void* stream) {
    if (n <= 0 || step <= 0) return SVLA_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, p_bf16, n, lr, beta1, beta2, eps,
                       bc1, bc2s, gnorm_sq, max_norm, grad_scale, 1.f - lr * weight_decay);
    return svla_launch_status();
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, bf16_t* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = f2bf(s[i]);
}
extern "C" int svla_cast_f32_bf16(const float* src, bf16_t* dst, long n, void* stream) {
    if (n <= 0) return SVLA_EINVAL;
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return svla_launch_status();
}

// dst[c, r] (bf16) = src[r, c] (f32): transposed bf16 weight copies for the input-gradient GEMMs
__global__ void transpose_cast_kernel(const float* __restrict__ s, int rows, int cols, bf16_t* __restrict__ d) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8)
        tile[j][tx] = (r0 + j < rows && c0 + tx < cols) ? s[(size_t)(r0 + j) * cols + c0 + tx] : 0.f;
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < cols && r0 + tx < rows) d[(size_t)(c0 + j) * rows + r0 + tx] = f2bf(tile[tx][j]);
}
extern "C" int svla_transpose_cast_f32_bf16(const float* src, int rows, int cols, bf16_t* dst, void* stream) {
    if (rows <= 0 || cols <= 0) return SVLA_EINVAL;
    hipLaunchKernelGGL(transpose_cast_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, rows,
                       cols, dst);
    return svla_launch_status();
}

// rows of a table -> bf16 rows (T5 shared embedding gather): out[i, :] = table[ids[i], :]
__device__ __forceinline__ void embed_gather_kernel_body(const float* __restrict__ table, const int64_t* __restrict__ ids, long n, int D,
                                    bf16_t* __restrict__ out) {
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n) return;
    const float* src = table + (size_t)ids[wave] * D;
    for (int c = lane * 2; c < D; c += 128) *(uint32_t*)(out + wave * D + c) = pack_bf2(src[c], src[c + 1]);
}
__global__ void embed_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids, long n, int D,
                                    bf16_t* __restrict__ out) { embed_gather_kernel_body(table, ids, n, D, out); }
extern "C" int svla_embed_gather_f32_bf16(const float* table, const int64_t* ids, long n, int D, bf16_t* out, void* stream) {
    if (n <= 0 || (D % 2)) return SVLA_EINVAL;
    SVLA_LAUNCH(embed_gather_kernel, embed_gather_kernel_body, 1024, 1, dim3((int)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, table, ids, n, D, out);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Order-independent 64-bit content hash of byte rows: h = sum_i (byte_i + 1) * mix(i).  One wave per row.
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void row_hash_kernel(const unsigned char* __restrict__ rows, long n_rows, int row_bytes, int64_t* __restrict__ out) {
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= n_rows) return;
    const unsigned char* p = rows + (size_t)wave * row_bytes;
    uint64_t h = 0;
    for (int i = lane; i < row_bytes; i += 64) h += (uint64_t)(p[i] + 1u) * splitmix64((uint64_t)i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)h, o, 64), hi = __shfl_xor((uint32_t)(h >> 32), o, 64);
        h += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) out[wave] = (int64_t)(h >> 1);  // keep it non-negative
}
extern "C" int svla_row_hash_u8(const unsigned char* rows, long n_rows, int row_bytes, int64_t* out, void* stream) {
    if (n_rows <= 0 || row_bytes <= 0) return SVLA_EINVAL;
    hipLaunchKernelGGL(row_hash_kernel, dim3((int)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, rows, n_rows, row_bytes, out);
    return svla_launch_status();
}

// dst[r, :] += src[r, :] for D-wide bf16 rows with independent row strides (token-0 rows of a [R, S, D] gradient)
__global__ void rows_add_kernel(bf16_t* __restrict__ dst, long dst_ld, const bf16_t* __restrict__ src, long src_ld, int rows, int D) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    for (int c = lane * 8; c < D; c += 512) {       // D = 512: one trip
        u32x4 a = *(const u32x4*)(dst + (size_t)wave * dst_ld + c);
        const u32x4 b = *(const u32x4*)(src + (size_t)wave * src_ld + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = pack_bf2(bf_lo(a[e]) + bf_lo(b[e]), bf_hi(a[e]) + bf_hi(b[e]));
        *(u32x4*)(dst + (size_t)wave * dst_ld + c) = a;
    }
}
extern "C" int svla_rows_add_bf16(bf16_t* dst, long dst_ld, const bf16_t* src, long src_ld, int rows, int D, void* stream) {
    if (rows <= 0 || D <= 0 || (D % 8) || (dst_ld % 8) || (src_ld % 8)) return SVLA_EINVAL;
    hipLaunchKernelGGL(rows_add_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dst, dst_ld, src, src_ld, rows, D);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Frozen-ViT preprocessor input (architecture/allenact_preprocessors/dino_preprocessors.py:224-239 normalise,
// :27-35 crop 3:-3 + 14x14/14 patch embedding): uint8 HWC frames -> normalised bf16 im2col rows [B, gh*gw, KP]
// with k = c*P*P + ky*P + kx (the flattened conv-weight order), zero padded to KP.  One block per (frame, patch row):
// the P image rows are staged in LDS with coalesced byte loads, outputs are written as coalesced dword pairs.
// Round 5: dword-wide.  The P image rows of the block are fetched as ALIGNED 4-byte words (256 B per wave instruction instead of 64; the crop starts
// 9 bytes into a row, so each staged row keeps its own 0..3-byte lead), the (c, ky, kx) decode of k is a per-block LDS table instead of three integer
// divisions per element, and every lane writes 16 bytes (8 bf16) of an output row: full 1-KiB wave stores.
__global__ void patchify_u8_kernel(const unsigned char* __restrict__ frames, int H, int W, int crop_x, int P, int gh, int gw,
                                   int KP, float m0, float m1, float m2, float s0, float s1, float s2, bf16_t* __restrict__ out, int wide) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / gh, gy = blockIdx.x % gh;
    const int rowbytes = gw * P * 3;
    const int pitch = (rowbytes + 8 + 3) & ~3;                  // staged row: up to 3 lead bytes + the segment, dword aligned
    unsigned char* rows = smem;                                // [P][pitch]
    unsigned short* lut = (unsigned short*)(smem + ((P * pitch + 15) & ~15));      // [KP]: byte offset of (ky, kx = 0.., c) inside the staged rows, patch column 0
    unsigned char* cls = (unsigned char*)(lut + KP);           // [KP]: channel of k (3 = padding)
    const unsigned char* src = frames + ((size_t)b * H + gy * P) * W * 3 + crop_x * 3;
    const int K = 3 * P * P;
    if (wide) {
        const int ndw = (rowbytes + 3 + 3) / 4;                 // upper bound of the words covering lead + segment of a row
        for (int i = threadIdx.x; i < P * ndw; i += blockDim.x) {
            const int r = i / ndw, j = i % ndw;
            const unsigned char* rp = src + (size_t)r * W * 3;
            const int lead = (int)((uintptr_t)rp & 3);
            // this row's own word count: the last word ends at most 3 bytes past the segment -- the slack the host's `wide` test proves (with the
            // uniform bound a row with lead 0 and rowbytes % 4 == 2 read 6 bytes past it: ADVICE r5)
            if (4 * j >= lead + rowbytes) continue;
            const uint32_t* ap = (const uint32_t*)((uintptr_t)rp & ~(uintptr_t)3);
            *(uint32_t*)(rows + r * pitch + 4 * j) = __builtin_nontemporal_load(ap + j);
        }
        for (int k = threadIdx.x; k < KP; k += blockDim.x) {
            if (k < K) {
                const int c = k / (P * P), ky = (k % (P * P)) / P, kx = k % P;
                const int lead = (int)((uintptr_t)(src + (size_t)ky * W * 3) & 3);
                lut[k] = (unsigned short)(ky * pitch + lead + kx * 3 + c);
                cls[k] = (unsigned char)c;
            } else { lut[k] = 0; cls[k] = 3; }
        }
    } else {
        for (int i = threadIdx.x; i < P * rowbytes; i += blockDim.x) rows[(i / rowbytes) * pitch + (i % rowbytes)] = src[(size_t)(i / rowbytes) * W * 3 + (i % rowbytes)];
        for (int k = threadIdx.x; k < KP; k += blockDim.x) {
            if (k < K) {
                const int c = k / (P * P), ky = (k % (P * P)) / P, kx = k % P;
                lut[k] = (unsigned short)(ky * pitch + kx * 3 + c);
                cls[k] = (unsigned char)c;
            } else { lut[k] = 0; cls[k] = 3; }
        }
    }
    __syncthreads();
    const float mean[4] = {m0, m1, m2, 0.f}, inv[4] = {1.f / s0, 1.f / s1, 1.f / s2, 0.f};
    bf16_t* dst = out + ((size_t)b * gh * gw + (size_t)gy * gw) * KP;
    const int kv = KP / 8;
    for (int i = threadIdx.x; i < gw * kv; i += blockDim.x) {
        const int gx = i / kv, k0 = (i % kv) * 8;
        const int xo = gx * P * 3;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cls[k0 + e];
            const float px = (float)rows[lut[k0 + e] + xo];
            v[e] = c < 3 ? (px / 255.0f - mean[c]) * inv[c] : 0.f;
        }
        u32x4 w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        *(u32x4*)(dst + (size_t)gx * KP + k0) = w;
    }
}
extern "C" int svla_patchify_u8_bf16(const unsigned char* frames, int B, int H, int W, int crop_x, int P, int gh, int gw, int KP,
                                     const float* mean3, const float* std3, bf16_t* out, void* stream) {
    if (B <= 0 || gh * P > H || crop_x + gw * P > W || KP < 3 * P * P || (KP % 8) || !mean3 || !std3) return SVLA_EINVAL;
    const int rowbytes = gw * P * 3, pitch = (rowbytes + 8 + 3) & ~3;
    if ((size_t)P * pitch + 64 > 60000) return SVLA_EINVAL;      // 16-bit offsets of the decode table
    const size_t lds = (((size_t)P * pitch + 15) & ~(size_t)15) + (size_t)KP * 3;
    // aligned word loads may touch up to 3 bytes on either side of a row segment: inside the frame buffer unless the crop ends at the very last byte of it
    const int wide = ((uintptr_t)frames % 4 == 0) && (crop_x > 0 || ((size_t)W * 3) % 4 == 0) && ((size_t)(crop_x + gw * P) * 3 + 3 <= (size_t)W * 3 || gh * P < H) ? 1 : 0;
    hipLaunchKernelGGL(patchify_u8_kernel, dim3(B * gh), dim3(256), lds, (hipStream_t)stream, frames, H, W, crop_x, P, gh, gw, KP,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out, wide);
    return svla_launch_status();
}

// DataAugmentationPreprocessor.process without augmentation (dino_preprocessors.py:224-239): u8 HWC -> (x/255 - mean)/std fp32 HWC
// Round 5: one aligned dword (4 bytes of the HWC stream) per lane -> one float4 per lane: 256-B wave loads, 1-KiB wave stores, non-temporal (stream-once).
__global__ void normalize_u8_kernel(const unsigned char* __restrict__ x, long n, float m0, float m1, float m2, float s0, float s1,
                                    float s2, float* __restrict__ y) {
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};      // (x / 255 - mean) / std exactly as the reference writes it
    const long nw = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (long)gridDim.x * blockDim.x) {
        const uint32_t w = __builtin_nontemporal_load((const uint32_t*)x + i);
        int c = (int)((4 * i) % 3);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = ((float)((w >> (8 * e)) & 0xffu) / 255.0f - mean[c]) / sd[c];
            c = c == 2 ? 0 : c + 1;
        }
        __builtin_nontemporal_store(o, (f32x4*)y + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {      // tail bytes (n % 4)
        const long i = (nw << 2) + threadIdx.x;
        const int c = (int)(i % 3);
        y[i] = ((float)x[i] / 255.0f - mean[c]) / sd[c];
    }
}
extern "C" int svla_normalize_u8_f32(const unsigned char* x, long n, const float* mean3, const float* std3, float* y, void* stream) {
    if (n <= 0 || (n % 3) || ((uintptr_t)x % 4) || ((uintptr_t)y % 16)) return SVLA_EINVAL;
    long blocks = ((n >> 2) + 255) / 256; if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(normalize_u8_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, n, mean3[0], mean3[1], mean3[2],
                       std3[0], std3[1], std3[2], y);
    return svla_launch_status();
}

// x_norm_patchtokens [B, skip + gh*gw, C] (bf16) -> AdaptiveAvgPool2d((oh, ow)) over the gh x gw patch grid
// (dino_preprocessors.py:31-35): writes bf16 tokens [B, ncam, oh*ow, C] slot ``cam`` and/or fp32 channels-first (B, C, oh, ow).
__global__ void adaptive_pool_kernel(const bf16_t* __restrict__ x, int skip, int gh, int gw, int C, int oh, int ow, int cam, int ncam,
                                     bf16_t* __restrict__ tok_out, float* __restrict__ chw_out) {
    const int b = blockIdx.x / (oh * ow), o = blockIdx.x % (oh * ow);
    const int oy = o / ow, ox = o % ow;
    const int y0 = (oy * gh) / oh, y1 = ((oy + 1) * gh + oh - 1) / oh;
    const int x0 = (ox * gw) / ow, x1 = ((ox + 1) * gw + ow - 1) / ow;
    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
    const bf16_t* src = x + (size_t)b * (skip + gh * gw) * C + (size_t)skip * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int yy = y0; yy < y1; ++yy)
            for (int xx = x0; xx < x1; ++xx) s += bf2f(src[(size_t)(yy * gw + xx) * C + c]);
        s *= inv;
        if (tok_out) tok_out[(((size_t)b * ncam + cam) * oh * ow + o) * C + c] = f2bf(s);
        if (chw_out) chw_out[(((size_t)b * C + c) * oh + oy) * ow + ox] = s;
    }
}
extern "C" int svla_adaptive_pool_tokens(const bf16_t* x, int B, int skip, int gh, int gw, int C, int oh, int ow, int cam, int ncam,
                                         bf16_t* tok_out, float* chw_out, void* stream) {
    if (B <= 0 || cam >= ncam || (!tok_out && !chw_out)) return SVLA_EINVAL;
    hipLaunchKernelGGL(adaptive_pool_kernel, dim3(B * oh * ow), dim3(128), 0, (hipStream_t)stream, x, skip, gh, gw, C, oh, ow, cam, ncam,
                       tok_out, chw_out);
    return svla_launch_status();
}

// y[b, 0, :] = cls + pos[0]; y[b, 1 + p, :] = patch[b, p, :] + pos[1 + p]   (DINOv2 prepare_tokens: cls token + position embedding)
// cls == NULL: no class token (SigLIP / timm ``class_token=False``, siglip_preprocessors.py:86-88): y[b, p, :] = patch[b, p, :] + pos[p].
__global__ void vit_tokens_kernel(const bf16_t* __restrict__ patch, const float* __restrict__ cls, const float* __restrict__ pos,
                                  int B, int NP, int C, bf16_t* __restrict__ y) {
    const int nc = cls ? 1 : 0, NT = NP + nc;
    const long n = (long)B * NT * (C / 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % (C / 2)) * 2;
        const long t = i / (C / 2);
        const int tok = (int)(t % NT);
        const long b = t / NT;
        float a0, a1;
        if (tok < nc) { a0 = cls[c]; a1 = cls[c + 1]; }
        else { const uint32_t w = *(const uint32_t*)(patch + ((size_t)b * NP + tok - nc) * C + c); a0 = bf_lo(w); a1 = bf_hi(w); }
        *(uint32_t*)(y + (size_t)t * C + c) = pack_bf2(a0 + pos[(size_t)tok * C + c], a1 + pos[(size_t)tok * C + c + 1]);
    }
}
extern "C" int svla_vit_tokens(const bf16_t* patch, const float* cls, const float* pos, int B, int NP, int C, bf16_t* y, void* stream) {
    if (B <= 0 || NP <= 0 || (C % 2)) return SVLA_EINVAL;
    long blocks = ((long)B * (NP + 1) * (C / 2) + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vit_tokens_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, patch, cls, pos, B, NP, C, y);
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------
// In-place dropout of a [rows, N] bf16 activation (N % 8 == 0) with the counter-based masks of include/svla.h: the two
// stand-alone sites of the frozen T5 encoder (after the token embedding and after the final layer norm; HF T5Stack), whose
// other dropouts ride in the GEMM / attention epilogues.
__device__ __forceinline__ void dropout_rows_kernel_body(bf16_t* __restrict__ x, long n8, int N, DropCfg drop) {
    drop = drop_resolve(drop);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        u32x4 w = *(u32x4*)(x + i * 8);
        const unsigned long long e0 = (unsigned long long)i * 8;       // row_mult == 1: flat index
        const unsigned keep = drop_keep4(drop, e0) | (drop_keep4(drop, e0 + 4) << 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float lo = ((keep >> (2 * e)) & 1u) ? bf_lo(w[e]) * drop.scale : 0.f;
            const float hi = ((keep >> (2 * e + 1)) & 1u) ? bf_hi(w[e]) * drop.scale : 0.f;
            w[e] = pack_bf2(lo, hi);
        }
        *(u32x4*)(x + i * 8) = w;
    }
}
__global__ void dropout_rows_kernel(bf16_t* __restrict__ x, long n8, int N, DropCfg drop) { dropout_rows_kernel_body(x, n8, N, drop); }
extern "C" int svla_dropout_bf16(bf16_t* x, long rows, int N, const svla_dropout* drop, void* stream) {
    if (rows <= 0 || N <= 0 || (N % 8)) return SVLA_EINVAL;
    const DropCfg c = drop_cfg(drop);
    if (!c.thr) return SVLA_OK;
    const long n8 = rows * N / 8;
    long blocks = (n8 + 255) / 256; if (blocks > 4096) blocks = 4096;
    SVLA_LAUNCH(dropout_rows_kernel, dropout_rows_kernel_body, 1024, 1, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, n8, N, c);
    return svla_launch_status();
}


// ------------------------------------------------------------------------------------------------
// llama KV-cache append of the acting path (llama/model.py:279-293: cache[:bsz, start_pos] = xk / xv): cache[b, t, :] = src[b, :] with
// the slot t read from DEVICE memory, so that a recorded / captured single-step launch sequence is step-independent.
__device__ __forceinline__ void kv_append_kernel_body(const bf16_t* __restrict__ src, long ld_src, bf16_t* __restrict__ cache, long cache_rows, int width,
                                 const int64_t* __restrict__ t_dev, int B) {
    const long t = *t_dev;
    const int cpr = width / 8;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * cpr; i += gridDim.x * blockDim.x) {
        const int b = i / cpr, c = (i % cpr) * 8;
        *(u32x4*)(cache + ((size_t)b * cache_rows + t) * width + c) = *(const u32x4*)(src + (size_t)b * ld_src + c);
    }
}
__global__ void kv_append_kernel(const bf16_t* __restrict__ src, long ld_src, bf16_t* __restrict__ cache, long cache_rows, int width,
                                 const int64_t* __restrict__ t_dev, int B) { kv_append_kernel_body(src, ld_src, cache, cache_rows, width, t_dev, B); }
extern "C" int svla_kv_append_bf16(const bf16_t* src, long ld_src, bf16_t* cache, long cache_rows, int width, const int64_t* t_dev, int B,
                                   void* stream) {
    if (B <= 0 || width <= 0 || (width % 8) || (ld_src % 8) || !t_dev) return SVLA_EINVAL;
    int blocks = (B * (width / 8) + 255) / 256; if (blocks > 1024) blocks = 1024;
    SVLA_LAUNCH(kv_append_kernel, kv_append_kernel_body, 1024, 1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, cache, cache_rows, width, t_dev, B);
    return svla_launch_status();
}

// ---- inputs of a recorded acting step -> the static buffers the recorded launches read (include/svla.h: svla_acting_stage) --------------------------
// One launch instead of ~20 framework copies / compares per env step: blocks [0, nb_tok) copy the DINO tokens (16 bytes per lane), block nb_tok + b stages env b:
// previous action, mask, object-in-hand, time step, goal ids (+ the T5 padding mask in both formats the kernels read) and row b of the KV-cache window mask
// kvalid[b, s] = (s <= t) & (s >= max(t - time_step[b], 0))  (allenact_dino_transformer.py:388-397); block nb_tok also writes the step counter and bumps the towers' dropout seeds.
struct ActingStageArgs {
    const u32x4* tok_src; u32x4* tok_dst; long n_tok16; int nb_tok;
    const int64_t* pa_src; int64_t* pa_dst; const float* mask_src; float* mask_dst; const int64_t* hand_src; int64_t* hand_dst;
    const int64_t* ts_src; int64_t* ts_dst; const int64_t* ids_src; int64_t* ids_dst; int64_t* am_dst; unsigned char* am8_dst;
    unsigned char* kvalid_dst; int64_t* t_dev; int B, L, max_steps, t; int* seed[3]; int seed_inc;
};
__global__ void acting_stage_kernel(ActingStageArgs a) {
    const int bid = blockIdx.x;
    if (bid < a.nb_tok) {
        for (long i = (long)bid * blockDim.x + threadIdx.x; i < a.n_tok16; i += (long)a.nb_tok * blockDim.x) a.tok_dst[i] = a.tok_src[i];
        return;
    }
    const int b = bid - a.nb_tok;
    const long ts = a.ts_src[b];
    if (threadIdx.x == 0) {
        a.pa_dst[b] = a.pa_src[b]; a.mask_dst[b] = a.mask_src[b]; a.hand_dst[b] = a.hand_src[b]; a.ts_dst[b] = ts;
        if (b == 0) {
            *a.t_dev = (int64_t)a.t;
#pragma unroll
            for (int k = 0; k < 3; ++k) if (a.seed[k]) *a.seed[k] = (int)((unsigned)*a.seed[k] + (unsigned)a.seed_inc);      // wraps in int32 like the framework's add_
        }
    }
    for (int j = threadIdx.x; j < a.L; j += blockDim.x) {
        const int64_t id = a.ids_src[(long)b * a.L + j];
        const int on = (id != 0 || j == 0) ? 1 : 0;
        a.ids_dst[(long)b * a.L + j] = id; a.am_dst[(long)b * a.L + j] = on; a.am8_dst[(long)b * a.L + j] = (unsigned char)on;
    }
    const long lo = (long)a.t - ts > 0 ? (long)a.t - ts : 0;
    for (int s0 = threadIdx.x; s0 < a.max_steps; s0 += blockDim.x) a.kvalid_dst[(long)b * a.max_steps + s0] = (unsigned char)((s0 <= a.t && s0 >= lo) ? 1 : 0);
}
extern "C" int svla_acting_stage(const void* tok_src, void* tok_dst, long tok_bytes, const int64_t* pa_src, int64_t* pa_dst, const float* mask_src, float* mask_dst,
                                 const int64_t* hand_src, int64_t* hand_dst, const int64_t* ts_src, int64_t* ts_dst, const int64_t* ids_src, int64_t* ids_dst,
                                 int64_t* am_dst, unsigned char* am8_dst, unsigned char* kvalid_dst, int64_t* t_dev, int B, int L, int max_steps, int t,
                                 int* seed0, int* seed1, int* seed2, int seed_inc, void* stream) {
    if (B <= 0 || L <= 0 || max_steps <= 0 || t < 0 || tok_bytes <= 0 || (tok_bytes % 16) || ((uintptr_t)tok_src % 16) || ((uintptr_t)tok_dst % 16)) return SVLA_EINVAL;
    if (!pa_src || !pa_dst || !mask_src || !mask_dst || !hand_src || !hand_dst || !ts_src || !ts_dst || !ids_src || !ids_dst || !am_dst || !am8_dst || !kvalid_dst || !t_dev) return SVLA_EINVAL;
    ActingStageArgs a;
    a.tok_src = (const u32x4*)tok_src; a.tok_dst = (u32x4*)tok_dst; a.n_tok16 = tok_bytes / 16;
    long nb = (a.n_tok16 + 1023) / 1024; if (nb > 512) nb = 512; if (nb < 1) nb = 1;
    a.nb_tok = (int)nb;
    a.pa_src = pa_src; a.pa_dst = pa_dst; a.mask_src = mask_src; a.mask_dst = mask_dst; a.hand_src = hand_src; a.hand_dst = hand_dst; a.ts_src = ts_src; a.ts_dst = ts_dst;
    a.ids_src = ids_src; a.ids_dst = ids_dst; a.am_dst = am_dst; a.am8_dst = am8_dst; a.kvalid_dst = kvalid_dst; a.t_dev = t_dev;
    a.B = B; a.L = L; a.max_steps = max_steps; a.t = t; a.seed[0] = seed0; a.seed[1] = seed1; a.seed[2] = seed2; a.seed_inc = seed_inc;
    hipLaunchKernelGGL(acting_stage_kernel, dim3(a.nb_tok + B), dim3(256), 0, (hipStream_t)stream, a);
    return svla_launch_status();
}

// ---- tower-grouped launches (csrc/launch.h; include/svla.h: svla_group_begin) -----------------------------------------------------------
static thread_local GroupCapture* t_group_open = nullptr;      // this thread's open capture
static thread_local GroupCapture* t_group_store = nullptr;     // allocated once per thread (~13 KiB of argument blocks)
GroupCapture* svla_group_capture() { return t_group_open; }
int svla_group_size() { return t_group_open ? t_group_open->size : 1; }
extern "C" int svla_group_begin(int members) {
    if (t_group_open || members < 1 || members > SVLA_MAXG) return SVLA_EINVAL;
    if (!t_group_store) t_group_store = new GroupCapture();
    GroupCapture* gc = t_group_store;
    gc->size = members;
    gc->member = 0;
    gc->overflow = 0;
    for (int m = 0; m < SVLA_MAXG; ++m) gc->n[m] = 0;
    t_group_open = gc;
    return SVLA_OK;
}
extern "C" int svla_group_member(int member) {
    GroupCapture* gc = t_group_open;
    if (!gc || member < 0 || member >= gc->size) return SVLA_EINVAL;
    gc->member = member;
    return SVLA_OK;
}
static bool group_same_launch(const DeferredLaunch& a, const DeferredLaunch& b) {
    return a.flush == b.flush && a.smem == b.smem && a.grid.x == b.grid.x && a.grid.y == b.grid.y && a.grid.z == 1 && b.grid.z == 1 &&
           a.block.x == b.block.x && a.block.y == b.block.y && a.block.z == b.block.z;
}
extern "C" int svla_group_end(void* stream) {
    GroupCapture* gc = t_group_open;
    if (!gc) return SVLA_EINVAL;
    t_group_open = nullptr;                      // the launches below are real
    hipStream_t st = (hipStream_t)stream;
    bool lockstep = !gc->overflow;
    for (int m = 1; m < gc->size; ++m) lockstep = lockstep && gc->n[m] == gc->n[0];
    int rc = SVLA_OK;
    if (lockstep) {
        for (int j = 0; j < gc->n[0] && rc == SVLA_OK; ++j) {
            const DeferredLaunch* ms[SVLA_MAXG];
            bool same = gc->size > 1;
            for (int m = 0; m < gc->size; ++m) {
                ms[m] = &gc->q[m][j];
                same = same && group_same_launch(gc->q[0][j], gc->q[m][j]);
            }
            if (same) { rc = ms[0]->flush(ms, gc->size, st); gc->grouped += 1; }
            else for (int m = 0; m < gc->size && rc == SVLA_OK; ++m) { rc = ms[m]->flush(&ms[m], 1, st); gc->single += 1; }
        }
    } else {        // the members did not issue the same number of launches (e.g. one of them has a ragged row tail): member by member, in call order
        for (int m = 0; m < gc->size && rc == SVLA_OK; ++m)
            for (int j = 0; j < gc->n[m] && rc == SVLA_OK; ++j) {
                const DeferredLaunch* one = &gc->q[m][j];
                rc = one->flush(&one, 1, st);
                gc->single += 1;
            }
    }
    return rc;
}
extern "C" int svla_group_stats(long* grouped, long* single) {
    GroupCapture* gc = t_group_store;
    if (grouped) *grouped = gc ? gc->grouped : 0;
    if (single) *single = gc ? gc->single : 0;
    if (gc) gc->grouped = gc->single = 0;
    return SVLA_OK;
}

// Zero a device buffer on the launch stream (gradient scratch that the text / embedding backward kernels accumulate into with atomics);
// a C-ABI entry so that it is part of a recorded launch sequence instead of a framework-side fill.
extern "C" int svla_zero_bytes(void* p, long bytes, void* stream) {
    if (!p || bytes <= 0) return SVLA_EINVAL;
    HIP_CHECK_RET(hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream));
    return SVLA_OK;
}
