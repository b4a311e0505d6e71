// fp32 kernels: (1) the small fp32 heads that are not MFMA-shaped (DiscreteCriticHead / MLPCriticHead of
// allenact_dino_transformer.py:720-766) and (2) the fp32-grade VERIFICATION MODE of the policy path (model.precision = "fp32"):
// the same kernel schedule as the bf16/MFMA product path with every activation, GEMM operand and attention probability in fp32,
// so that logits / values / losses / lambda can be compared with the reference's fp32 arithmetic at fp32 tolerance (the reference
// runs everything in fp32, SURVEY 8: "All reference arithmetic is fp32").  Correct and simple, not fast: LDS-tiled FMA GEMM,
// one workgroup per (row, head) attention with the score matrix in LDS, wave-per-row norms.  Dropout uses the same counter-based
// masks as the bf16 kernels (include/svla.h: svla_dropout), so train-mode runs are comparable too.
#include "common.h"

// ------------------------------------------------------------------------------------------------ strided GEMM
// C[m, n] (ld = ldc) = epi( alpha * sum_k A(m,k) * B(n,k) ),  A(m,k) = A[m*sam + k*sak],  B(n,k) = B[n*sbn + k*sbk]
//   epi(v) = v + bias[n] -> act (0 none, 1 ReLU, 2 GELU-erf) -> dropout -> zero where mask[m*ldm + n] <= 0 -> + residual[m*ldr + n]
//   accumulate != 0: C += epi(...)   (weight gradients accumulate into the fp32 gradient arena)
// Any of NT / NN / TN is a choice of strides.  64 x 64 tile, 16-deep K steps through LDS, 4 x 4 outputs per thread.
struct GemmF32Args {
    const float* A; long sam, sak;
    const float* B; long sbn, sbk;
    const float* bias; const float* residual; long ldr; const float* mask; long ldm;
    float* C; long ldc;
    int M, N, K, act, accumulate;
    float alpha;
    DropCfg drop;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmF32Args p) {
    p.drop = drop_resolve(p.drop);
    __shared__ float As[16][64 + 1], Bs[16][64 + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = tid + 256 * e;           // 1024 elements of each 64 x 16 operand tile
            // pick the thread -> element map that walks the operand's unit-stride dimension fastest
            int ra, ka, rb, kb;
            if (p.sak == 1) { ka = q & 15; ra = q >> 4; } else { ra = q & 63; ka = q >> 6; }
            if (p.sbk == 1) { kb = q & 15; rb = q >> 4; } else { rb = q & 63; kb = q >> 6; }
            const int m = m0 + ra, n = n0 + rb;
            As[ka][ra] = (m < p.M && k0 + ka < p.K) ? p.A[(size_t)m * p.sam + (size_t)(k0 + ka) * p.sak] : 0.f;
            Bs[kb][rb] = (n < p.N && k0 + kb < p.K) ? p.B[(size_t)n * p.sbn + (size_t)(k0 + kb) * p.sbk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            float v = acc[i][j] * p.alpha + (p.bias ? p.bias[n] : 0.f);
            if (p.act == 1) v = fmaxf(v, 0.f);
            else if (p.act == 2) v = gelu_erf(v);
            if (p.drop.thr) {
                const unsigned long long e = (unsigned long long)m * p.drop.row_mult * p.N + n;
                const unsigned keep = drop_keep4(p.drop, e & ~3ull);
                v = ((keep >> (e & 3)) & 1u) ? v * p.drop.scale : 0.f;
            }
            if (p.mask && !(p.mask[(size_t)m * p.ldm + n] > 0.f)) v = 0.f;
            if (p.residual) v += p.residual[(size_t)m * p.ldr + n];
            float* c = p.C + (size_t)m * p.ldc + n;
            *c = p.accumulate ? *c + v : v;
        }
    }
}

extern "C" int svla_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk, const float* bias,
                             const float* residual, long ldr, const float* mask, long ldm, float* C, long ldc, int M, int N, int K,
                             int act, int accumulate, float alpha, const svla_dropout* drop, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || act < 0 || act > 2 || !A || !B || !C) return SVLA_EINVAL;
    GemmF32Args p{A, sam, sak, B, sbn, sbk, bias, residual, ldr, mask, ldm, C, ldc, M, N, K, act, accumulate, alpha, drop_cfg(drop)};
    hipLaunchKernelGGL(gemm_f32_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, p);
    return svla_launch_status();
}

// out[n] += sum_m X[m*row_stride*ldx + n]  (bias gradients), one block per 64 columns, rows strided over the block's 4 waves
__global__ void colsum_f32_kernel(const float* __restrict__ X, long ldx, int M, int N, int row_stride, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (n < N)
        for (int m = blockIdx.y * 4 + w; m < M; m += gridDim.y * 4) s += X[(size_t)m * row_stride * ldx + n];
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && n < N) atomicAdd(out + n, red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

extern "C" int svla_colsum_f32(const float* X, long ldx, int M, int N, int row_stride, float* out, void* stream) {
    if (M <= 0 || N <= 0) return SVLA_EINVAL;
    int gy = (M + 255) / 256; if (gy > 256) gy = 256;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3((N + 63) / 64, gy), dim3(256), 0, (hipStream_t)stream, X, ldx, M, N, row_stride > 0 ? row_stride : 1, out);
    return svla_launch_status();
}

// ================================================================================================ attention (fp32)
// Same contract as svla_attn_fwd_bf16 / svla_attn_bwd_bf16 (include/svla.h) with fp32 tensors: one workgroup per (batch row, head),
// one wave per query (forward, dQ) or per key (dK/dV); scores, softmax and both contractions in fp32.  LSE = natural-log
// log-sum-exp of the scaled, biased, masked scores.
struct AttnF32Args {
    const float *Q, *K, *V; long ld;
    float* O; long ldo;
    float* LSE;
    const float* dO; long lddo;
    float *dQ, *dK, *dV; long ldd;
    const int* traj; const float* bias; const unsigned char* kvalid;
    int S, H, mask_mode; float scale; int kv_rows, Sq; long ldq, lddq;
    DropCfg drop;
    int hd;           // head width: 64 everywhere in the policy; 96 in two imitation-learning presets (TransformerConfig(n, 768, 8)); <= 128, lane l owns dims l and l + 64
};
#define AF_MAXS 512
#define AF_KPL (AF_MAXS / 64)
#define AF_MAXHD 128

__device__ __forceinline__ bool af_masked(const AttnF32Args& p, int r, int q, int k) {
    if (p.mask_mode == 1 && (k > q || p.traj[(size_t)r * p.S + k] != p.traj[(size_t)r * p.S + q])) return true;
    if (p.kvalid && !p.kvalid[(size_t)r * p.S + k]) return true;
    return false;
}
__device__ __forceinline__ bool af_keep(const AttnF32Args& p, int r, int h, int q, int k) {
    if (!p.drop.thr) return true;
    const unsigned long long e = ((unsigned long long)((size_t)r * p.H + h) * p.S + q) * (unsigned long long)((p.S + 3) & ~3) + k;
    const unsigned x = drop_bits(p.drop.key, e >> 1);
    return ((e & 1) ? (x >> 16) : (x & 0xffffu)) >= p.drop.thr;
}
__device__ __forceinline__ float af_dot(const float* a, const float* b, int hd) {
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < hd; ++d) s = fmaf(a[d], b[d], s);
    return s;
}
// scaled + biased score of (q, k), or -inf when masked
__device__ __forceinline__ float af_score(const AttnF32Args& p, int r, int h, int q, int k, const float* qrow, const float* krow) {
    if (af_masked(p, r, q, k)) return -INFINITY;
    float s = af_dot(qrow, krow, p.hd) * p.scale;
    if (p.bias) s += p.bias[((size_t)h * p.S + q) * p.S + k];
    return s;
}

__global__ void __launch_bounds__(256) attn_fwd_f32_kernel(AttnF32Args p) {
    p.drop = drop_resolve(p.drop);
    __shared__ float ps[4][AF_MAXS];
    __shared__ float qs[4][AF_MAXHD];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = blockIdx.x / p.H, h = blockIdx.x % p.H, hd = p.hd;
    const bool hi = lane + 64 < hd;           // this lane also owns dim lane + 64
    const float* Kb = p.K + (size_t)r * p.kv_rows * p.ld + h * hd;
    const float* Vb = p.V + (size_t)r * p.kv_rows * p.ld + h * hd;
    for (int q = wid; q < p.Sq; q += 4) {
        const size_t qtok = (size_t)r * p.Sq + q;
        if (lane < hd) qs[wid][lane] = p.Q[qtok * p.ldq + h * hd + lane];
        if (hi) qs[wid][lane + 64] = p.Q[qtok * p.ldq + h * hd + lane + 64];
        __builtin_amdgcn_wave_barrier();
        float sc[AF_KPL], mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < AF_KPL; ++u) {
            const int k = lane + 64 * u;
            sc[u] = k < p.S ? af_score(p, r, h, q, k, qs[wid], Kb + (size_t)k * p.ld) : -INFINITY;
            mx = fmaxf(mx, sc[u]);
        }
        mx = wave_max(mx);
        float se = 0.f;
#pragma unroll
        for (int u = 0; u < AF_KPL; ++u) { sc[u] = (sc[u] == -INFINITY) ? 0.f : expf(sc[u] - mx); se += sc[u]; }
        se = wave_sum(se);
        const float inv = se > 0.f ? 1.f / se : 0.f;
        if (p.LSE && lane == 0) p.LSE[((size_t)r * p.H + h) * p.Sq + q] = se > 0.f ? mx + logf(se) : -INFINITY;
#pragma unroll
        for (int u = 0; u < AF_KPL; ++u) {
            const int k = lane + 64 * u;
            if (k < p.S) ps[wid][k] = af_keep(p, r, h, q, k) ? sc[u] * inv * p.drop.scale : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        float o = 0.f, o2 = 0.f;
        for (int k = 0; k < p.S; ++k) {
            const float pk = ps[wid][k];
            if (lane < hd) o = fmaf(pk, Vb[(size_t)k * p.ld + lane], o);
            if (hi) o2 = fmaf(pk, Vb[(size_t)k * p.ld + lane + 64], o2);
        }
        if (lane < hd) p.O[qtok * p.ldo + h * hd + lane] = o;
        if (hi) p.O[qtok * p.ldo + h * hd + lane + 64] = o2;
        __builtin_amdgcn_wave_barrier();
    }
}

// phase 1: dQ per query (and D_q = rowsum(dO * O) into LDS); phase 2: dK / dV per key, recomputing the probabilities
__global__ void __launch_bounds__(256) attn_bwd_f32_kernel(AttnF32Args p) {
    p.drop = drop_resolve(p.drop);
    __shared__ float ws[4][AF_MAXS], ws2[4][AF_MAXS];
    __shared__ float Dq[AF_MAXS];
    __shared__ float vec[4][AF_MAXHD], vec2[4][AF_MAXHD];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = blockIdx.x / p.H, h = blockIdx.x % p.H, hd = p.hd;
    const bool lo = lane < hd, hi = lane + 64 < hd;
    const float* Kb = p.K + (size_t)r * p.S * p.ld + h * hd;
    const float* Vb = p.V + (size_t)r * p.S * p.ld + h * hd;
    for (int q = wid; q < p.Sq; q += 4) {
        const size_t qtok = (size_t)r * p.Sq + q;
        float dsum = 0.f;
        if (lo) {
            vec[wid][lane] = p.Q[qtok * p.ldq + h * hd + lane];
            const float dov = p.dO[qtok * p.lddo + h * hd + lane];
            vec2[wid][lane] = dov;
            dsum = dov * p.O[qtok * p.ldo + h * hd + lane];
        }
        if (hi) {
            vec[wid][lane + 64] = p.Q[qtok * p.ldq + h * hd + lane + 64];
            const float dov = p.dO[qtok * p.lddo + h * hd + lane + 64];
            vec2[wid][lane + 64] = dov;
            dsum += dov * p.O[qtok * p.ldo + h * hd + lane + 64];
        }
        const float D = wave_sum(dsum);
        if (lane == 0) Dq[q] = D;
        __builtin_amdgcn_wave_barrier();
        const float lse = p.LSE[((size_t)r * p.H + h) * p.Sq + q];
#pragma unroll
        for (int u = 0; u < AF_KPL; ++u) {
            const int k = lane + 64 * u;
            if (k < p.S) {
                const float s = af_score(p, r, h, q, k, vec[wid], Kb + (size_t)k * p.ld);
                const float pr = (s == -INFINITY) ? 0.f : expf(s - lse);
                float dp = af_dot(vec2[wid], Vb + (size_t)k * p.ld, hd);
                dp = af_keep(p, r, h, q, k) ? dp * p.drop.scale : 0.f;
                ws[wid][k] = pr * (dp - D) * p.scale;           // dS (scaled)
            }
        }
        __builtin_amdgcn_wave_barrier();
        float dq = 0.f, dq2 = 0.f;
        for (int k = 0; k < p.S; ++k) {
            const float w = ws[wid][k];
            if (lo) dq = fmaf(w, Kb[(size_t)k * p.ld + lane], dq);
            if (hi) dq2 = fmaf(w, Kb[(size_t)k * p.ld + lane + 64], dq2);
        }
        if (lo) p.dQ[qtok * p.lddq + h * hd + lane] = dq;
        if (hi) p.dQ[qtok * p.lddq + h * hd + lane + 64] = dq2;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int k = wid; k < p.S; k += 4) {
        if (lo) { vec[wid][lane] = Kb[(size_t)k * p.ld + lane]; vec2[wid][lane] = Vb[(size_t)k * p.ld + lane]; }
        if (hi) { vec[wid][lane + 64] = Kb[(size_t)k * p.ld + lane + 64]; vec2[wid][lane + 64] = Vb[(size_t)k * p.ld + lane + 64]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < AF_KPL; ++u) {
            const int q = lane + 64 * u;
            if (q < p.Sq) {
                const size_t qtok = (size_t)r * p.Sq + q;
                const float s = af_score(p, r, h, q, k, p.Q + qtok * p.ldq + h * hd, vec[wid]);
                const float pr = (s == -INFINITY) ? 0.f : expf(s - p.LSE[((size_t)r * p.H + h) * p.Sq + q]);
                float dp = af_dot(p.dO + qtok * p.lddo + h * hd, vec2[wid], hd);
                const bool keep = af_keep(p, r, h, q, k);
                dp = keep ? dp * p.drop.scale : 0.f;
                ws[wid][q] = pr * (dp - Dq[q]) * p.scale;       // dS[q, k]
                ws2[wid][q] = keep ? pr * p.drop.scale : 0.f;   // dropped-out probability
            }
        }
        __builtin_amdgcn_wave_barrier();
        float dk = 0.f, dv = 0.f, dk2 = 0.f, dv2 = 0.f;
        for (int q = 0; q < p.Sq; ++q) {
            const size_t qtok = (size_t)r * p.Sq + q;
            const float a = ws[wid][q], b = ws2[wid][q];
            if (lo) { dk = fmaf(a, p.Q[qtok * p.ldq + h * hd + lane], dk); dv = fmaf(b, p.dO[qtok * p.lddo + h * hd + lane], dv); }
            if (hi) { dk2 = fmaf(a, p.Q[qtok * p.ldq + h * hd + lane + 64], dk2); dv2 = fmaf(b, p.dO[qtok * p.lddo + h * hd + lane + 64], dv2); }
        }
        const size_t ktok = (size_t)r * p.S + k;
        if (lo) { p.dK[ktok * p.ldd + h * hd + lane] = dk; p.dV[ktok * p.ldd + h * hd + lane] = dv; }
        if (hi) { p.dK[ktok * p.ldd + h * hd + lane + 64] = dk2; p.dV[ktok * p.ldd + h * hd + lane + 64] = dv2; }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int svla_attn_fwd_f32(const float* Q, const float* K, const float* V, long ld, float* O, long ldo, float* LSE, int rows, int S,
                                 int H, int head_dim, float scale, int mask_mode, const int* traj, const float* bias,
                                 const unsigned char* kvalid, int Sq, long ldq, int kv_rows, const svla_dropout* drop, void* stream) {
    if (rows <= 0 || S <= 0 || S > AF_MAXS || H <= 0 || head_dim <= 0 || head_dim > AF_MAXHD || (mask_mode == 1 && !traj) || Sq < 0 || Sq > S) return SVLA_EINVAL;
    if (kv_rows && kv_rows < S) return SVLA_EINVAL;
    AttnF32Args p{Q, K, V, ld, O, ldo, LSE, nullptr, 0, nullptr, nullptr, nullptr, 0, traj, bias, kvalid, S, H, mask_mode, scale,
                  kv_rows ? kv_rows : S, Sq ? Sq : S, Sq ? ldq : ld, 0, drop_cfg(drop), head_dim};
    hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3(rows * H), dim3(256), 0, (hipStream_t)stream, p);
    return svla_launch_status();
}

extern "C" int svla_attn_bwd_f32(const float* Q, const float* K, const float* V, long ld, const float* O, long ldo, const float* LSE,
                                 const float* dO, long lddo, float* dQ, float* dK, float* dV, long ldd, int rows, int S, int H,
                                 int head_dim, float scale, int mask_mode, const int* traj, const unsigned char* kvalid, int Sq, long ldq,
                                 long lddq, const svla_dropout* drop, void* stream) {
    if (rows <= 0 || S <= 0 || S > AF_MAXS || H <= 0 || head_dim <= 0 || head_dim > AF_MAXHD || (mask_mode == 1 && !traj) || Sq < 0 || Sq > S || !LSE) return SVLA_EINVAL;
    AttnF32Args p{Q, K, V, ld, const_cast<float*>(O), ldo, const_cast<float*>(LSE), dO, lddo, dQ, dK, dV, ldd, traj, nullptr, kvalid, S, H,
                  mask_mode, scale, S, Sq ? Sq : S, Sq ? ldq : ld, Sq ? lddq : ldd, drop_cfg(drop), head_dim};
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3(rows * H), dim3(256), 0, (hipStream_t)stream, p);
    return svla_launch_status();
}

// ================================================================================================ glue (fp32 twins of misc.hip)
// Same arithmetic and argument meaning as the bf16 entry points of the same name; activations are fp32, D-wide rows.
__global__ void feat_to_tokens_f32_kernel(const float* __restrict__ feat, int R, int C, int P, int cam, int ncam, float* __restrict__ out) {
    const long n = (long)R * C * P;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int pp = (int)(i % P), c = (int)((i / P) % C);
        const long r = i / ((long)P * C);
        out[((r * ncam + cam) * P + pp) * C + c] = feat[i];
    }
}
extern "C" int svla_feat_to_tokens_f32(const float* feat, int R, int C, int P, int cam, int ncam, float* out, void* stream) {
    if (R <= 0 || C <= 0 || P <= 0 || cam >= ncam) return SVLA_EINVAL;
    long blocks = ((long)R * C * P + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(feat_to_tokens_f32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, feat, R, C, P, cam, ncam, out);
    return svla_launch_status();
}

__global__ void fusion_fill_f32_kernel(const float* __restrict__ fusion_token, const float* __restrict__ text, const int* __restrict__ gid,
                                       int R, int S, int L, int text_off, int D, float* __restrict__ x0) {
    const int r = blockIdx.x;
    float* row = x0 + (size_t)r * S * D;
    const float* tsrc = text + (size_t)gid[r] * L * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        row[c] = fusion_token[c];
        for (int j = 0; j < L; ++j) row[(size_t)(text_off + j) * D + c] = tsrc[(size_t)j * D + c];
    }
}
extern "C" int svla_fusion_fill_f32(const float* fusion_token, const float* text, const int* gid, int R, int S, int L, int text_off,
                                    int D, float* x0, void* stream) {
    if (R <= 0 || text_off + L > S || D <= 0) return SVLA_EINVAL;
    hipLaunchKernelGGL(fusion_fill_f32_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, fusion_token, text, gid, R, S, L, text_off, D, x0);
    return svla_launch_status();
}

__global__ void fusion_text_bwd_f32_kernel(const float* __restrict__ dx0, const int* __restrict__ gid, int T, int B, int S, int L,
                                           int text_off, int D, float* __restrict__ dtext) {
    const int b = blockIdx.x, j = blockIdx.y;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float acc = 0.f;
        int cur = gid[b];
        for (int t = 0; t < T; ++t) {
            const int r = t * B + b, g = gid[r];
            if (g != cur) { atomicAdd(dtext + ((size_t)cur * L + j) * D + c, acc); acc = 0.f; cur = g; }
            acc += dx0[((size_t)r * S + text_off + j) * D + c];
        }
        atomicAdd(dtext + ((size_t)cur * L + j) * D + c, acc);
    }
}
extern "C" int svla_fusion_text_bwd_f32(const float* dx0, const int* gid, int T, int B, int S, int L, int text_off, int D, float* dtext, void* stream) {
    if (T <= 0 || B <= 0 || L <= 0 || D <= 0) return SVLA_EINVAL;
    hipLaunchKernelGGL(fusion_text_bwd_f32_kernel, dim3(B, L), dim3(256), 0, (hipStream_t)stream, dx0, gid, T, B, S, L, text_off, D, dtext);
    return svla_launch_status();
}

__global__ void decoder_embed_f32_kernel(const float* __restrict__ xf, long xf_row_stride, const float* __restrict__ act_tab,
                                         const float* __restrict__ hand_tab, const float* __restrict__ div_term,
                                         const int64_t* __restrict__ prev_actions, const float* __restrict__ masks,
                                         const int64_t* __restrict__ hand, const int64_t* __restrict__ time_step, int T, int B, int n_actions,
                                         int D, float* __restrict__ out) {
    const int row = blockIdx.x;            // (t*B + b)
    const int t = row / B, b = row % B;
    const int64_t a = masks[row] != 0.f ? prev_actions[row] : (int64_t)n_actions;
    const float pos = (float)time_step[row];
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float ang = pos * div_term[c >> 1];
        const float pe = (c & 1) ? cosf(ang) : sinf(ang);
        // reference order: time_enc + ((obs + prev_action_emb) + in_hand_emb)
        out[((size_t)b * T + t) * D + c] = pe + ((xf[(size_t)row * xf_row_stride + c] + act_tab[(size_t)a * D + c]) + hand_tab[(size_t)hand[row] * D + c]);
    }
}
extern "C" int svla_decoder_embed_fwd_f32(const float* xf, long xf_row_stride, const float* act_tab, const float* hand_tab,
                                          const float* div_term, const int64_t* prev_actions, const float* masks, const int64_t* hand,
                                          const int64_t* time_step, int T, int B, int n_actions, int D, float* out, void* stream) {
    if (T <= 0 || B <= 0 || D <= 0) return SVLA_EINVAL;
    hipLaunchKernelGGL(decoder_embed_f32_kernel, dim3(T * B), dim3(256), 0, (hipStream_t)stream, xf, xf_row_stride, act_tab, hand_tab, div_term,
                       prev_actions, masks, hand, time_step, T, B, n_actions, D, out);
    return svla_launch_status();
}

__global__ void decoder_embed_bwd_f32_kernel(const float* __restrict__ dout, const int64_t* __restrict__ prev_actions,
                                             const float* __restrict__ masks, const int64_t* __restrict__ hand, int T, int B, int n_actions, int D,
                                             float* __restrict__ dxf, long dxf_row_stride, float* __restrict__ d_act_tab,
                                             float* __restrict__ d_hand_tab) {
    const int row = blockIdx.x;
    const int t = row / B, b = row % B;
    const int a = masks[row] != 0.f ? (int)prev_actions[row] : n_actions;
    const int hh = (int)hand[row];
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float v = dout[((size_t)b * T + t) * D + c];
        dxf[(size_t)row * dxf_row_stride + c] = v;
        atomicAdd(d_act_tab + (size_t)a * D + c, v);
        atomicAdd(d_hand_tab + (size_t)hh * D + c, v);
    }
}
extern "C" int svla_decoder_embed_bwd_f32(const float* dout, const int64_t* prev_actions, const float* masks, const int64_t* hand, int T,
                                          int B, int n_actions, int D, float* dxf, long dxf_row_stride, float* d_act_tab, float* d_hand_tab,
                                          void* stream) {
    if (T <= 0 || B <= 0 || D <= 0) return SVLA_EINVAL;
    hipLaunchKernelGGL(decoder_embed_bwd_f32_kernel, dim3(T * B), dim3(256), 0, (hipStream_t)stream, dout, prev_actions, masks, hand, T, B,
                       n_actions, D, dxf, dxf_row_stride, d_act_tab, d_hand_tab);
    return svla_launch_status();
}

__global__ void rows_add_f32_kernel(float* __restrict__ dst, long dst_ld, const float* __restrict__ src, long src_ld, int rows, int D) {
    const long n = (long)rows * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[(i / D) * dst_ld + (i % D)] += src[(i / D) * src_ld + (i % D)];
}
extern "C" int svla_rows_add_f32(float* dst, long dst_ld, const float* src, long src_ld, int rows, int D, void* stream) {
    if (rows <= 0 || D <= 0) return SVLA_EINVAL;
    long blocks = ((long)rows * D + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rows_add_f32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, dst, dst_ld, src, src_ld, rows, D);
    return svla_launch_status();
}

__global__ void swiglu_fwd_f32_kernel(const float* __restrict__ ab, long M, int Hd, float* __restrict__ g) {
    const long n = M * Hd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / Hd; const int c = (int)(i % Hd);
        const float a = ab[m * 2 * Hd + c], b = ab[m * 2 * Hd + Hd + c];
        g[i] = a / (1.f + expf(-a)) * b;
    }
}
__global__ void swiglu_bwd_f32_kernel(const float* __restrict__ ab, const float* __restrict__ dg, long M, int Hd, float* __restrict__ dab) {
    const long n = M * Hd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / Hd; const int c = (int)(i % Hd);
        const float a = ab[m * 2 * Hd + c], b = ab[m * 2 * Hd + Hd + c], d = dg[i];
        const float sg = 1.f / (1.f + expf(-a));
        dab[m * 2 * Hd + c] = d * b * sg * (1.f + a * (1.f - sg));
        dab[m * 2 * Hd + Hd + c] = d * a * sg;
    }
}
extern "C" int svla_swiglu_fwd_f32(const float* ab, long M, int Hd, float* g, void* stream) {
    if (M <= 0 || Hd <= 0) return SVLA_EINVAL;
    long blocks = (M * Hd + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(swiglu_fwd_f32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, ab, M, Hd, g);
    return svla_launch_status();
}
extern "C" int svla_swiglu_bwd_f32(const float* ab, const float* dg, long M, int Hd, float* dab, void* stream) {
    if (M <= 0 || Hd <= 0) return SVLA_EINVAL;
    long blocks = (M * Hd + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(swiglu_bwd_f32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, ab, dg, M, Hd, dab);
    return svla_launch_status();
}

__global__ void embed_gather_f32_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids, long n, int D, float* __restrict__ out) {
    const long tot = n * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long)gridDim.x * blockDim.x)
        out[i] = table[(size_t)ids[i / D] * D + (i % D)];
}
extern "C" int svla_embed_gather_f32(const float* table, const int64_t* ids, long n, int D, float* out, void* stream) {
    if (n <= 0 || D <= 0) return SVLA_EINVAL;
    long blocks = (n * D + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(embed_gather_f32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, table, ids, n, D, out);
    return svla_launch_status();
}

__global__ void dropout_f32_kernel(float* __restrict__ x, long rows, int N, DropCfg drop) {
    drop = drop_resolve(drop);
    const long n4 = rows * N / 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const unsigned keep = drop_keep4(drop, (unsigned long long)i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) x[i * 4 + e] = ((keep >> e) & 1u) ? x[i * 4 + e] * drop.scale : 0.f;
    }
}
extern "C" int svla_dropout_f32(float* x, long rows, int N, const svla_dropout* drop, void* stream) {
    if (rows <= 0 || N <= 0 || (N % 4)) return SVLA_EINVAL;
    const DropCfg c = drop_cfg(drop);
    if (!c.thr) return SVLA_OK;
    long blocks = (rows * N / 4 + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(dropout_f32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, N, c);
    return svla_launch_status();
}
