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
