// bf16 MFMA GEMMs for the linear layers of the policy (gfx950, v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   svla_gemm_nt_bf16 : C[M,N] = epi(A[M,K] . B[N,K]^T)   forward linears and, with pre-transposed weights,
//                       the input-gradient GEMMs (dX = dY . W).  Epilogue: +bias[n], ReLU/GELU, ReLU-mask from a
//                       saved activation, +residual, bf16 or fp32 output.
//   svla_gemm_tn_f32acc: dW[N,K] += sum_m dY[m,n] X[m,k]   weight-gradient GEMM: reduction over the (huge) row
//                       dimension, split across workgroups, fp32 atomics into the fp32 gradient buffer.
//
// Shapes on this path: M = rows x tokens (1e5..2e6), N,K in {384,512,1536,2048}: A streams from HBM once, the
// weights stay L2-resident.  Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32 tiles.
// Register-staged double-buffered LDS pipeline (one barrier per K-tile), XOR-swizzled 16-byte chunks so both the
// ds_write_b128 staging stores and the ds_read_b128 fragment loads are bank-conflict free, XCD-aware tile order so
// the N-tiles that share an A panel run back-to-back on one XCD's L2, LDS-staged epilogue with 16-byte stores.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256

// physical 16-byte chunk inside a 128-byte (64 x bf16) LDS row
__device__ __forceinline__ int swz_nt(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// bijective XCD remap: workgroup b runs on XCD b % 8; give every XCD a contiguous range of tile ids.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

struct GemmNtArgs {
    const bf16_t* A; long lda;
    const bf16_t* B; long ldb;
    const float* bias;
    const bf16_t* residual; long ldr;
    const bf16_t* relu_mask; long ldm;   // zero outputs where relu_mask <= 0 (input-gradient of ReLU)
    void* C; long ldc;
    int M, N, K, act, out_f32;
    float alpha;
};

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

__global__ void __launch_bounds__(NTHREADS, 2) gemm_nt_bf16_kernel(GemmNtArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* As = (bf16_t*)smem;                 // [2][BM][BK]
    bf16_t* Bs = As + 2 * BM * BK;              // [2][BN][BK]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int ntn = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    // staging assignment: 4 chunks of A and 4 of B per thread per K-tile
    int srow[4], schunk[4];
    const bf16_t* ga[4];
    const bf16_t* gb[4];
    bool aok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = tid + NTHREADS * j;
        srow[j] = q >> 3; schunk[j] = q & 7;
        aok[j] = (m0 + srow[j]) < p.M;
        ga[j] = p.A + (size_t)(aok[j] ? m0 + srow[j] : 0) * p.lda + schunk[j] * 8;
        gb[j] = p.B + (size_t)(n0 + srow[j]) * p.ldb + schunk[j] * 8;
    }
    u32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = aok[j] ? *(const u32x4*)(ga[j] + k0) : u32x4{0, 0, 0, 0};
            rb[j] = *(const u32x4*)(gb[j] + k0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = srow[j] * BK + swz_nt(srow[j], schunk[j]) * 8;
            *(u32x4*)(As + buf * BM * BK + off) = ra[j];
            *(u32x4*)(Bs + buf * BN * BK + off) = rb[j];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // epilogue operands of this thread's 8 output chunks (row = (tid + 256 j) >> 4, 16-byte chunk ecc = tid & 15)
    const int ecc = tid & 15;
    float ebias[8];
    u32x4 eres[8], emask[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ebias[e] = p.bias ? p.bias[n0 + ecc * 8 + e] : 0.f;
    if (!p.out_f32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = m0 + ((tid + NTHREADS * j) >> 4);
            eres[j] = (p.residual && m < p.M) ? *(const u32x4*)(p.residual + (size_t)m * p.ldr + n0 + ecc * 8) : u32x4{0, 0, 0, 0};
            emask[j] = (p.relu_mask && m < p.M) ? *(const u32x4*)(p.relu_mask + (size_t)m * p.ldm + n0 + ecc * 8) : u32x4{0, 0, 0, 0};
        }
    }

    const int nk = p.K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const bf16_t* Ab = As + buf * BM * BK;
        const bf16_t* Bb = Bs + buf * BN * BK;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ra_ = wm * 64 + t * 32 + fr, rb_ = wn * 64 + t * 32 + fr;
                fa[t] = *(const bf16x8*)(Ab + ra_ * BK + swz_nt(ra_, kk * 2 + fh) * 8);
                fb[t] = *(const bf16x8*)(Bb + rb_ * BK + swz_nt(rb_, kk * 2 + fh) * 8);
            }
            // operands swapped (W rows as the MFMA "A" side): each lane then owns 4 consecutive output columns
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma32(fb[j], fa[i], acc[i][j]);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j][reg]: output row m = wm*64 + i*32 + (lane&31),
    //                                 output col n = wn*64 + j*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (p.out_f32) {
        float* C = (float*)p.C;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + i * 32 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = n0 + wn * 64 + j * 32 + 8 * rg + 4 * fh;
                    float4 o;
                    float* op = (float*)&o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][rg * 4 + e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f);
                        if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                        else if (p.act == ACT_GELU) v = gelu_f(v);
                        if (p.relu_mask && !(bf2f(p.relu_mask[(size_t)m * p.ldm + n + e]) > 0.f)) v = 0.f;
                        if (p.residual) v += bf2f(p.residual[(size_t)m * p.ldr + n + e]);
                        op[e] = v;
                    }
                    *(float4*)(C + (size_t)m * p.ldc + n) = o;
                }
        }
        return;
    }
    // fp32 tile staged through LDS (reusing the operand buffers: all MFMA reads are behind the loop's last barrier);
    // bias / activation / ReLU-mask / residual are applied in fp32 on whole 16-byte output chunks whose mask and
    // residual operands were prefetched before the K loop (their HBM latency hides under the main loop).
    constexpr int CS = BN + 4;  // padded fp32 row: 528 B, conflict-free ds_write_b128 / ds_read_b128
    float* Cs = (float*)smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int nl = wn * 64 + j * 32 + 8 * rg + 4 * fh;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e] * p.alpha;
                *(f32x4*)(Cs + (wm * 64 + i * 32 + fr) * CS + nl) = v;
            }
    __syncthreads();
    bf16_t* C = (bf16_t*)p.C;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = tid + NTHREADS * j;
        const int row = q >> 4;
        const int m = m0 + row;
        if (m >= p.M) continue;
        const f32x4 a = *(const f32x4*)(Cs + row * CS + ecc * 8), b = *(const f32x4*)(Cs + row * CS + ecc * 8 + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float lo = v[2 * e] + ebias[2 * e], hi = v[2 * e + 1] + ebias[2 * e + 1];
            if (p.act == ACT_RELU) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            else if (p.act == ACT_GELU) { lo = gelu_f(lo); hi = gelu_f(hi); }
            if (p.relu_mask) {
                if (!(bf_lo(emask[j][e]) > 0.f)) lo = 0.f;
                if (!(bf_hi(emask[j][e]) > 0.f)) hi = 0.f;
            }
            if (p.residual) { lo += bf_lo(eres[j][e]); hi += bf_hi(eres[j][e]); }
            w[e] = pack_bf2(lo, hi);
        }
        *(u32x4*)(C + (size_t)m * p.ldc + n0 + ecc * 8) = w;
    }
}

extern "C" int svla_gemm_nt_bf16(const bf16_t* A, long lda, const bf16_t* B, long ldb, const float* bias,
                                 const bf16_t* residual, long ldr, const bf16_t* relu_mask, long ldm, void* C, long ldc,
                                 int M, int N, int K, int act, int out_f32, float alpha, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (N % BN) || (K % BK)) return SVLA_EINVAL;
    if ((lda % 8) || (ldb % 8) || (ldc % (out_f32 ? 4 : 8)) || (residual && (ldr % 8)) || (relu_mask && (ldm % 8))) return SVLA_EINVAL;
    GemmNtArgs p{A, lda, B, ldb, bias, residual, ldr, relu_mask, ldm, C, ldc, M, N, K, act, out_f32, alpha};
    const int mt = (M + BM - 1) / BM, nt = N / BN;
    const size_t lds = BM * (BN + 4) * sizeof(float);  // 66 KiB: max(operand double buffers 64 KiB, fp32 epilogue tile)
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_nt_bf16_kernel, dim3(mt * nt), dim3(NTHREADS), lds, (hipStream_t)stream, p);
    return svla_launch_status();
}

// =================================================================================================
// Weight-gradient GEMM:  dW[N,K] += sum_{m in chunk} dY[m,n] * X[m,k]      (both operands "transposed":
// the reduction index m is the slow memory dimension).  LDS tiles stay row-major [64 m][128 cols]; the MFMA
// fragments (8 reduction slots for one output row/col) are gathered with ds_read_b64_tr_b16.
#define TK 64  // reduction rows per LDS tile
__device__ __forceinline__ int swz_tn(int row, int chunk) { return chunk ^ ((row & 3) << 2); }

struct GemmTnArgs {
    const bf16_t* dY; long ldy;   // [M, N]
    const bf16_t* X; long ldx;    // [M, K]
    float* dW; long ldw;          // [N, K] fp32, accumulated with atomics
    int M, N, K, chunk_rows;
};

__device__ __forceinline__ bf16x8 frag_tr(const bf16_t* tile, int step, int col0, int lane) {
    // 8 reduction slots (tile rows step*16 + 8*(lane>>5) + 0..7) for column col0 + (lane&31)
    const int p = lane & 15, q = lane >> 4;
    const int colq = col0 + 16 * (q & 1) + 4 * (p & 3);          // first of the 4 columns this lane addresses
    const int r0 = step * 16 + 8 * (q >> 1) + (p >> 2);
    const int r1 = r0 + 4;
    const bf16x4 lo = lds_tr16_b64(tile + r0 * 128 + swz_tn(r0, colq >> 3) * 8 + (colq & 7));
    const bf16x4 hi = lds_tr16_b64(tile + r1 * 128 + swz_tn(r1, colq >> 3) * 8 + (colq & 7));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ void __launch_bounds__(NTHREADS, 2) gemm_tn_bf16_kernel(GemmTnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ys = (bf16_t*)smem;            // [2][TK][128]
    bf16_t* Xs = Ys + 2 * TK * 128;        // [2][TK][128]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid >> 1, wk = wid & 1;
    const int ntk = p.K / 128, ntn = p.N / 128;
    const int ntile = ntn * ntk;
    const int tile = blockIdx.x % ntile, chunk = blockIdx.x / ntile;
    const int n0 = (tile / ntk) * 128, k0 = (tile % ntk) * 128;
    const int mbeg = chunk * p.chunk_rows;
    const int mend = min(p.M, mbeg + p.chunk_rows);

    int srow[4], schunk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int q = tid + NTHREADS * j; srow[j] = q >> 4; schunk[j] = q & 15; }
    u32x4 ry[4], rx[4];
    auto gload = [&](int mb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = mb + srow[j];
            const bool ok = m < mend;
            ry[j] = ok ? *(const u32x4*)(p.dY + (size_t)m * p.ldy + n0 + schunk[j] * 8) : u32x4{0, 0, 0, 0};
            rx[j] = ok ? *(const u32x4*)(p.X + (size_t)m * p.ldx + k0 + schunk[j] * 8) : u32x4{0, 0, 0, 0};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = srow[j] * 128 + swz_tn(srow[j], schunk[j]) * 8;
            *(u32x4*)(Ys + buf * TK * 128 + off) = ry[j];
            *(u32x4*)(Xs + buf * TK * 128 + off) = rx[j];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (mend - mbeg + TK - 1) / TK;
    if (nt <= 0) return;
    gload(mbeg);
    lstore(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(mbeg + (t + 1) * TK);
        const bf16_t* Yb = Ys + buf * TK * 128;
        const bf16_t* Xb = Xs + buf * TK * 128;
#pragma unroll
        for (int s = 0; s < TK / 16; ++s) {
            bf16x8 fy[2], fx[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                fy[u] = frag_tr(Yb, s, wn * 64 + u * 32, lane);
                fx[u] = frag_tr(Xb, s, wk * 64 + u * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma32(fy[i], fx[j], acc[i][j]);
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }
    // acc[i][j][reg]: n = n0 + wn*64 + i*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5);  k = k0 + wk*64 + j*32 + (lane&31)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int k = k0 + wk * 64 + j * 32 + (lane & 31);
                atomicAdd(p.dW + (size_t)n * p.ldw + k, acc[i][j][r]);
            }
}

extern "C" int svla_gemm_tn_f32acc(const bf16_t* dY, long ldy, const bf16_t* X, long ldx, float* dW, long ldw, int M, int N,
                                   int K, void* stream) {
    if (M <= 0 || (N % 128) || (K % 128) || (ldy % 8) || (ldx % 8)) return SVLA_EINVAL;
    const int ntile = (N / 128) * (K / 128);
    // aim for ~2048 workgroups; chunk is a multiple of the 64-row reduction tile
    int chunks = (2048 + ntile - 1) / ntile;
    int chunk_rows = ((M + chunks - 1) / chunks + TK - 1) / TK * TK;
    if (chunk_rows < 4 * TK) chunk_rows = 4 * TK;
    chunks = (M + chunk_rows - 1) / chunk_rows;
    GemmTnArgs p{dY, ldy, X, ldx, dW, ldw, M, N, K, chunk_rows};
    const size_t lds = 2 * 2 * TK * 128 * sizeof(bf16_t);  // 64 KiB
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(ntile * chunks), dim3(NTHREADS), lds, (hipStream_t)stream, p);
    return svla_launch_status();
}

// Column sums (bias gradients): db[n] += sum_m dY[m, n].  HBM-bound single pass, 16-byte loads.
__global__ void colsum_bf16_kernel(const bf16_t* __restrict__ dY, long ldy, int M, int N, int row_stride_groups,
                                   float* __restrict__ db) {
    // each thread owns 8 consecutive columns; blockDim.x threads cover N columns (N/8 <= blockDim.x) x rows/block
    const int cpr = N / 8;                       // chunks per row
    const int rows_per_pass = blockDim.x / cpr;
    const int c = threadIdx.x % cpr, rl = threadIdx.x / cpr;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rl < rows_per_pass) {
        for (long m = (long)blockIdx.x * rows_per_pass + rl; m < M; m += (long)gridDim.x * rows_per_pass) {
            const u32x4 w = *(const u32x4*)(dY + (size_t)m * row_stride_groups * ldy + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[2 * e] += bf_lo(w[e]); s[2 * e + 1] += bf_hi(w[e]); }
        }
    }
    extern __shared__ float red[];  // [blockDim.x][8]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
    __syncthreads();
    for (int col = threadIdx.x; col < N; col += blockDim.x) {
        const int cc = col >> 3, e = col & 7;
        float t = 0.f;
        for (int r = 0; r < rows_per_pass; ++r) t += red[(r * cpr + cc) * 8 + e];
        atomicAdd(&db[col], t);
    }
}

// row m is read at memory row m*row_stride (row_stride > 1: e.g. token 0 of every [S, D] group)
extern "C" int svla_colsum_bf16(const bf16_t* dY, long ldy, int M, int N, int row_stride, float* db, void* stream) {
    if (M <= 0 || N <= 0 || (N % 8) || N / 8 > 256 || (ldy % 8)) return SVLA_EINVAL;
    const int threads = 256;
    const int rpp = threads / (N / 8);
    int blocks = (M + rpp - 1) / rpp;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(blocks), dim3(threads), threads * 8 * sizeof(float), (hipStream_t)stream, dY, ldy,
                       M, N, row_stride > 0 ? row_stride : 1, db);
    return svla_launch_status();
}
