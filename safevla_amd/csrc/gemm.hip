// bf16 MFMA GEMMs for the linear layers of the policy (gfx950, v_mfma_f32_32x32x16_bf16, fp32 accumulate).
//
//   svla_gemm_nt_bf16 : C[M,N] = epi(alpha * A[M,K] . B[N,K]^T)   forward linears and, with pre-transposed weights, the
//                       input-gradient GEMMs (dX = dY . W).  Epilogue: +bias[n], ReLU/GELU, train-mode dropout, ReLU mask (bf16
//                       activation or 1-bit), +residual, bf16 or fp32 output, optional ReLU sign-bit output.
//   svla_gemm_tn_f32acc: dW[N,K] += sum_m dY[m,n] X[m,k]   weight-gradient GEMM: reduction over the (huge) row dimension split
//                       across workgroups, fp32 atomics into the fp32 gradient arena, fused bias gradient.
//
// Shapes on this path: M = rows x tokens (1e5..2e6), N,K in {384,512,1536,2048}: A streams from HBM once, the weights stay
// L2-resident.  Four kernels:
//   gemm_nt256k64_bf16_kernel : persistent 256x256x64 tile, 8 waves -- every big row-streaming NT GEMM (the roofline kernel)
//   gemm_nt_bf16_kernel       : 128x128x32 tile, 4 waves, 4-stage LDS-DMA pipeline -- small M*N, fp32 outputs
//   gemm_tn256_bf16_kernel    : 256x256 output tile, 64-row stages, transposed ds_read_b64_tr_b16 fragments -- big dW
//   gemm_tn_bf16_kernel       : 128x128 output tile, register-staged -- small dW, K % 256 != 0
// Common idioms: global_load_lds (LDS-DMA) staging with the XOR swizzle applied on the DMA SOURCE address (the DMA writes
// lane-linear) and on the fragment ds_reads (bank-conflict free, measured), counted s_waitcnt vmcnt + raw s_barrier, XCD-aware
// tile order so the tiles that share an operand panel run on one XCD's L2.
#include "common.h"
#include "asm_kernels.h"
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <mutex>

#define BM 128
#define BN 128
#define BK 32
#define NST 4     // LDS pipeline stages (3 K-tiles of DMA in flight)
#define NTHREADS 256

// physical 16-byte chunk inside a 64-byte (32 x bf16) LDS row: 16 consecutive rows x one logical chunk hit 16 distinct
// 16-byte bank slots (conflict-free ds_read_b128)
__device__ __forceinline__ int swz_nt(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

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
    unsigned char* bits_out;          // act == ReLU: also write the output's sign bits (blocked layout, see relu_bits_word)
    const unsigned char* bits_in;     // ReLU mask given as such bits instead of a bf16 activation tensor (relu_mask)
    DropCfg drop;                     // train-mode dropout after the activation, before the residual add (thr == 0: off)
    int row0;         // global row index of A's row 0 (dropout counter / sign-bit block of the M-tail sub-problem behind the assembly kernels); 128-tile kernel only
    int dbg;          // timing-only ablations of the 256-tile kernel (tools/ab_gemm.py, tools/shape_gemm.py): 1 = no C stores,
                      // 2 = no epilogue, 64 = no fragment reads / MFMAs (operand DMA stream + barriers alone)
    float rms_eps;    // > 0 (svla_gemm_nt_rmsa_bf16, 128-tile kernel only): row m of the product is scaled by rsqrt(mean_k A[m,k]^2 + rms_eps) -- RMSNorm(A) . W^T
                      // with the norm's gamma folded into W; the row sums of squares fall out of the A fragments the MFMAs read anyway
};

// ReLU sign bits live in a kernel-private blocked layout: [ceil(M/32)][N/64][32 rows][8 bytes]: the 64 bits of (row m, 64-column
// group) are one 8-byte word and a 32-row x 64-column slab (one wave's epilogue unit in the 256-tile kernel) is 256 contiguous bytes
__device__ __forceinline__ size_t relu_bits_word(int m, int n, int N) { return ((size_t)(m >> 5) * (N >> 6) + (n >> 6)) * 256 + (size_t)(m & 31) * 8; }

__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) {
    typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// erf-GELU of nn.GELU / timm's Mlp as x * (1/2 + clamp(t P(z))) with t = clamp(x, +-4.5), z = t^2 * (2 / 20.25) - 1 and a degree-9 polynomial P (asmgen/
// gelu_poly.py: |error| <= 2.2e-5 for all x, exact tails): 15 plain VALU operations, no transcendental -- round 4's Abramowitz-Stegun form was 17 + v_rcp +
// v_exp, and the exposed GELU epilogue was 40 % of the ViT's fc1 GEMM.  The assembly GELU flavour (svla_nt_as_k384_f2) evaluates the same polynomial.
#include "_obj/gelu_poly.h"
__device__ __forceinline__ float gelu_f(float x) {
    constexpr float c[SVLA_GELU_DEGREE + 1] = SVLA_GELU_COEFS;
    const float t = __builtin_amdgcn_fmed3f(x, -SVLA_GELU_CLAMP, SVLA_GELU_CLAMP);
    const float z = fmaf(t * t, SVLA_GELU_ZSCALE, -1.f);
    float p = c[SVLA_GELU_DEGREE];
#pragma unroll
    for (int k = SVLA_GELU_DEGREE - 1; k >= 0; --k) p = fmaf(p, z, c[k]);
    const float g = __builtin_amdgcn_fmed3f(t * p, -0.5f, 0.5f);
    return x * (g + 0.5f);
}

__device__ __forceinline__ void gemm_nt_bf16_kernel_body(GemmNtArgs p) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* As = (bf16_t*)smem;                 // [NST][BM][BK]
    bf16_t* Bs = As + NST * BM * BK;            // [NST][BN][BK]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int ntn = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    // staging: global -> LDS DMA (global_load_lds_dwordx4, 1 KiB = 16 tile rows of 64 B per wave instruction; 2 A + 2 B
    // per wave per K-tile).  The DMA writes lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address:
    // lane i lands at (row rbase + i/4, physical chunk i%4) and therefore fetches logical chunk (i%4) ^ swz(row).
    const bf16_t* ga[2];
    const bf16_t* gb[2];
    int ldsoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rbase = (wid * 2 + j) * 16;
        const int row = rbase + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        const int am = min(m0 + row, p.M - 1);          // M tail: clamp (those output rows are never stored)
        ga[j] = p.A + (size_t)am * p.lda + c * 8;
        gb[j] = p.B + (size_t)(n0 + row) * p.ldb + c * 8;
        ldsoff[j] = __builtin_amdgcn_readfirstlane(rbase * BK);
    }
    auto stage = [&](int st, int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[j] + k0),
                                             (__attribute__((address_space(3))) void*)(As + st * BM * BK + ldsoff[j]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[j] + k0),
                                             (__attribute__((address_space(3))) void*)(Bs + st * BN * BK + ldsoff[j]), 16, 0, 0);
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
    for (int e = 0; e < 8; ++e) ebias[e] = 0.f;
    if (p.bias) {
        const f32x4 b0 = *(const f32x4*)(p.bias + n0 + ecc * 8), b1 = *(const f32x4*)(p.bias + n0 + ecc * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ebias[e] = b0[e]; ebias[4 + e] = b1[e]; }
    }
    // branch-free per lane (rows past M are clamped: never stored); only wave-uniform branches on the pointers
#pragma unroll
    for (int j = 0; j < 8; ++j) { eres[j] = u32x4{0, 0, 0, 0}; emask[j] = u32x4{0, 0, 0, 0}; }
    if (!p.out_f32 && p.residual) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = min(m0 + ((tid + NTHREADS * j) >> 4), p.M - 1);
            eres[j] = *(const u32x4*)(p.residual + (size_t)m * p.ldr + n0 + ecc * 8);
        }
    }
    if (!p.out_f32 && p.relu_mask) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = min(m0 + ((tid + NTHREADS * j) >> 4), p.M - 1);
            emask[j] = *(const u32x4*)(p.relu_mask + (size_t)m * p.ldm + n0 + ecc * 8);
        }
    }
    unsigned ebits = 0;       // 8 rows x 8 mask bits
    unsigned ebits_hi = 0;
    if (!p.out_f32 && p.bits_in) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = min(m0 + ((tid + NTHREADS * j) >> 4), p.M - 1);
            const unsigned b = p.bits_in[relu_bits_word(m, n0 + ecc * 8, p.N) + (((n0 + ecc * 8) & 63) >> 3)];
            if (j < 4) ebits |= b << (8 * j); else ebits_hi |= b << (8 * (j - 4));
        }
    }

    // NST-stage pipeline: up to NST-1 K-tiles of DMA in flight; ONE raw s_barrier per K-tile.  Each wave waits (counted
    // vmcnt, 4 DMA instructions per tile per wave) for its own pieces of tile kt, the barrier then publishes the whole
    // tile and proves every wave is done reading the stage the next DMA overwrites.
    const int nk = p.K / BK;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s * BK);
    const int fr = lane & 31, fh = lane >> 5;
    float ssq[2] = {0.f, 0.f};          // rms_eps > 0: sum of squares of A rows wm*64 + t*32 + fr over this lane's half of every 16-wide k-step
    for (int kt = 0; kt < nk; ++kt) {
        const int rem = nk - 1 - kt;       // tiles issued after tile kt that may stay in flight
        if (rem >= NST - 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NST - 1 < nk) stage((kt + NST - 1) % NST, (kt + NST - 1) * BK);
        const int st = kt % NST;
        const bf16_t* Ab = As + st * BM * BK;
        const bf16_t* Bb = Bs + st * BN * BK;
        // all fragment reads of the tile first (one exposed LDS latency per tile), then 8 back-to-back MFMAs
        bf16x8 fa[BK / 16][2], fb[BK / 16][2];
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ra_ = wm * 64 + t * 32 + fr, rb_ = wn * 64 + t * 32 + fr;
                fa[kk][t] = *(const bf16x8*)(Ab + ra_ * BK + swz_nt(ra_, kk * 2 + fh) * 8);
                fb[kk][t] = *(const bf16x8*)(Bb + rb_ * BK + swz_nt(rb_, kk * 2 + fh) * 8);
            }
        // operands swapped (W rows as the MFMA "A" side): each lane then owns 4 consecutive output columns
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma32(fb[kk][j], fa[kk][i], acc[i][j]);
        if (p.rms_eps > 0.f) {             // wave-uniform
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float a = bf2f((bf16_t)fa[kk][t][e]); ssq[t] = fmaf(a, a, ssq[t]); }
        }
    }
    float rrow[2] = {1.f, 1.f};         // the accumulators of block row i belong to A row wm*64 + i*32 + fr: the row this lane summed
    if (p.rms_eps > 0.f) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float s = ssq[t] + __shfl_xor(ssq[t], 32, 64);      // the two k-halves of a step sit in lanes fr and fr + 32
            rrow[t] = rsqrtf(s / (float)p.K + p.rms_eps);
        }
    }
    __syncthreads();   // all MFMA reads of the operand stages are done before the epilogue reuses the LDS

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
                        float v = acc[i][j][rg * 4 + e] * (p.alpha * rrow[i]) + (p.bias ? p.bias[n + e] : 0.f);
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
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rg * 4 + e] * (p.alpha * rrow[i]);
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
        unsigned obits = 0, dkeep = 0;
        if (p.drop.thr) {
            const unsigned long long e0 = (unsigned long long)(m + p.row0) * p.drop.row_mult * p.N + n0 + ecc * 8;
            dkeep = drop_keep4(p.drop, e0) | (drop_keep4(p.drop, e0 + 4) << 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float lo = v[2 * e] + ebias[2 * e], hi = v[2 * e + 1] + ebias[2 * e + 1];
            if (p.act == ACT_RELU) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
            else if (p.act == ACT_GELU) { lo = gelu_f(lo); hi = gelu_f(hi); }
            if (p.drop.thr) {
                lo = ((dkeep >> (2 * e)) & 1u) ? lo * p.drop.scale : 0.f;
                hi = ((dkeep >> (2 * e + 1)) & 1u) ? hi * p.drop.scale : 0.f;
            }
            if (p.relu_mask) {
                if (!(bf_lo(emask[j][e]) > 0.f)) lo = 0.f;
                if (!(bf_hi(emask[j][e]) > 0.f)) hi = 0.f;
            }
            if (p.bits_in) {
                const unsigned b = ((j < 4 ? ebits : ebits_hi) >> (8 * (j & 3))) & 0xffu;
                if (!((b >> (2 * e)) & 1u)) lo = 0.f;
                if (!((b >> (2 * e + 1)) & 1u)) hi = 0.f;
            }
            if (p.residual) { lo += bf_lo(eres[j][e]); hi += bf_hi(eres[j][e]); }
            w[e] = pack_bf2(lo, hi);
            if (lo > 0.f) obits |= 1u << (2 * e);
            if (hi > 0.f) obits |= 2u << (2 * e);
        }
        *(u32x4*)(C + (size_t)m * p.ldc + n0 + ecc * 8) = w;
        if (p.bits_out) p.bits_out[relu_bits_word(m, n0 + ecc * 8, p.N) + (((n0 + ecc * 8) & 63) >> 3)] = (unsigned char)obits;
    }
}
__global__ void __launch_bounds__(NTHREADS, 2) gemm_nt_bf16_kernel(GemmNtArgs p) { gemm_nt_bf16_kernel_body(p); }

// =================================================================================================
// 256x256 tile kernel (8 waves = 2(M) x 4(N), wave tile 128x64 = 4x2 MFMA 32x32 tiles) for the big row-streaming GEMMs.
// Why 256x256: a 128x128 tile needs 32 KiB of operands per 64-deep K step per 512 MFMA cycles = the CU's whole L1->LDS
// path (measured: 47 % of wave time parked, 24 % MFMA busy); 256x256 halves the bytes per FLOP and the LDS reads per MFMA
// (6 fragments per 8 MFMAs).
// PERSISTENT: one workgroup per CU walks a list of tiles; the K-tiles of consecutive tiles form one continuous LDS-DMA
// stream, so the first K-tile of the next output tile is in flight while the current tile's epilogue runs.
#define NT256_THREADS 512
// BK = 64: LDS rows are 128 bytes, so every LDS-DMA instruction moves 8 FULL 128-byte lines (a BK = 32 image fetches half
// lines, 16 rows x 64 B per instruction, and the other half of each line one K-step later: measured -11 % at K = 2048,
// -30 % on 8192^3).  NS64 = 2 operand buffers of 64 KiB + 8 wave-private 4 KiB epilogue buffers = 160 KiB of LDS.
// Epilogue: bias/activation/mask/residual in fp32 on the accumulator layout, one bf16 rounding, then each 32 x 64 slab is
// transposed through the wave's LDS buffer so that every global store instruction writes 8 full 128-byte lines.
#define BK64 64
#define NS64 2
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// Epilogue of the 256x256 kernels (shared by the persistent 2-buffer kernel and the 8-phase kernel): bias / activation / dropout /
// mask / residual in fp32 on the accumulator layout acc[i][j][reg] (row = wm*128 + i*32 + (lane&31), col = wn*64 + j*32 + (reg&3) +
// 8*(reg>>2) + 4*(lane>>5)), one bf16 rounding, then each 32 x 64 slab is transposed through a wave-private 4 KiB LDS buffer so
// that every global store instruction writes 8 full 128-byte lines.
// The epilogue is VALU-issue bound (measured with in-kernel cycle counters: ~4 cycles per vector instruction per SIMD, two waves per
// SIMD), so its cost is its instruction count: addresses are a wave-uniform base + one lane offset (no 64-bit vector arithmetic), ReLU
// and the sign bits work on the packed bf16 pairs (v_pk_max_i16 / v_pk_min_i16), the dropout hash of a slab shares its row products
// (2 instead of ~4.6 quarter-rate v_mul_lo_u32 per hash, bit-identical masks), dropout is a compile-time flavour (no per-group branch).
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned pk_min_i16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int ACT, int AUX, int HI_OFF, bool LDS_BIAS, bool DROP>
struct Nt256Epi {
    static constexpr bool HAS_RES = (AUX & 1) != 0, HAS_MASK = (AUX & 2) != 0, HAS_BITS = AUX == 4, LDS_AUX = HAS_RES || HAS_MASK;
    static constexpr bool PACKED_RELU = ACT == ACT_RELU && AUX == 0;      // nothing is added after the activation: ReLU commutes with the rounding
    const GemmNtArgs& p;
    const int wm, wn, lane;
    const char* aux;
    long ldaux;
    // wave-private staging of one 32 x 64 bf16 slab (128-byte rows): write (row fr, 8-byte piece), read (row lane>>3 [+8 it], 16-byte
    // chunk lane&7).  Rows 0-15 start at Es, rows 16-31 at Es + hi_off (2048: one contiguous 4-KiB buffer; 16384: two 2-KiB stripes
    // of neighbouring ring slots, see gemm_nt8p)
    char* Es;
    static constexpr int hi_off = HI_OFF;
    const float* bias_lds;    // LDS-resident copy of bias[0..N) (8-phase kernel: no vector-memory load in the bias path), or nullptr
    // the residual / mask slab is fetched row-major (8 full 128-byte lines per instruction), one slab ahead, and turned into the
    // accumulator layout through the wave's LDS buffer (the inverse of the output transposition)
    u32x4 auxrm[3][4];      // slab i lives in buffer i % 3: loaded TWO slabs ahead (one slab of epilogue work is ~1 k cycles, an HBM access ~2 k)
    // AUX == 4: the ReLU mask as 1 bit per element ([M, N/8] bytes): the 64 bits of this lane's slab row are ONE 8-byte load in
    // the accumulator layout -- no LDS round trip, 16x fewer mask bytes than a bf16 activation tensor
    u32x2 mbits[2];
    __device__ __forceinline__ Nt256Epi(const GemmNtArgs& p_, char* es_wave, const float* bias_lds_, int wid, int lane_)
        : p(p_), wm(__builtin_amdgcn_readfirstlane(wid >> 2)), wn(__builtin_amdgcn_readfirstlane(wid & 3)), lane(lane_) {
        aux = (const char*)(HAS_RES ? p.residual : (HAS_MASK ? p.relu_mask : nullptr));
        ldaux = HAS_RES ? p.ldr : p.ldm;
        Es = es_wave;
        bias_lds = bias_lds_;
    }
    __device__ __forceinline__ int it_off(int it) const { return (it & 1) * 1024 + (it >> 1) * hi_off; }     // rows it*8 .. it*8+7
    // lo: an opaque copy of the lane id made where the epilogue starts -- everything lane-dependent is derived from it there, otherwise
    // hipcc hoists ~35 VGPRs of address arithmetic across the main loop
    __device__ __forceinline__ void load_aux(int i, u32x4 (&buf)[4], int tm0, int tn0, int lo) {
        const uint32_t loff = (uint32_t)(((lo >> 3) * ldaux + (lo & 7) * 8) * 2);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r0 = tm0 + wm * 128 + i * 32 + it * 8;      // wave-uniform first row of this 8-row piece
            if (r0 + 8 <= p.M) buf[it] = *(const u32x4*)(aux + ((size_t)r0 * ldaux + tn0 + wn * 64) * 2 + loff);
            else buf[it] = *(const u32x4*)(aux + ((size_t)min(r0 + (lo >> 3), p.M - 1) * ldaux + tn0 + wn * 64 + (lo & 7) * 8) * 2);      // M tail: clamp (never stored)
        }
    }
    __device__ __forceinline__ void load_bits(int i, u32x2& dst, int tm0, int tn0, int lo) {
        const int mc = min(tm0 + wm * 128 + i * 32 + (lo & 31), p.M - 1);
        dst = *(const u32x2*)(p.bits_in + relu_bits_word(mc, tn0 + wn * 64, p.N));
    }
    // slab 0's operands: issued while the last K-tile is still being computed
    __device__ __forceinline__ void prefetch0(int m0, int n0) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        if (LDS_AUX && n0 + wn * 64 < p.N) { load_aux(0, auxrm[0], m0, n0, lo); load_aux(1, auxrm[1], m0, n0, lo); }
        if (HAS_BITS && n0 + wn * 64 < p.N) load_bits(0, mbits[0], m0, n0, lo);
    }
    // returns whether this wave issued exactly 16 (+4 sign-bit) stores (full tile, columns inside N)
    __device__ __forceinline__ bool run(f32x16 (&acc)[4][2], int m0, int n0) {
        char* C = (char*)p.C;
        const bool wave_cols_valid = n0 + wn * 64 < p.N;      // wave-uniform: a wave's 64 output columns are all inside N or all outside
        if ((p.dbg & 2) || !wave_cols_valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
            return false;
        }
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int fr = lo & 31, fh = lo >> 5;
        const int e_wr = ((fr & 15) * 128 + (fr >> 4) * hi_off + fh * 8) | ((fr & 7) << 4);      // ^ (group << 4): the 8-byte piece of group (j, rg)
        const int e_rd = (lo >> 3) * 128 + (((lo & 7) ^ (lo >> 3)) << 4);
        const uint32_t c_loff = (uint32_t)(((lo >> 3) * p.ldc + (lo & 7) * 8) * 2);
        f32x4 bias4[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                bias4[j][rg] = LDS_BIAS ? *(const f32x4*)(bias_lds + n0 + wn * 64 + j * 32 + 8 * rg + 4 * fh)
                                        : (p.bias ? *(const f32x4*)(p.bias + n0 + wn * 64 + j * 32 + 8 * rg + 4 * fh) : f32x4{0.f, 0.f, 0.f, 0.f});
        // one slab in three steps: convert(i) -> 8 packed 8-byte pieces (+ sign bits); write them into the staging buffer; read the slab back
        // row-major and store it.  Without an LDS-staged operand (no residual / mask tensor) the steps of consecutive slabs are software-
        // pipelined: slab i+1 is converted while slab i's read-back is in flight (LDS operations of one wave execute in order, so the one
        // 4-KiB buffer is enough) -- the LDS round trip was ~300 exposed cycles per slab.
        auto convert = [&](const int i, u32x2 (&pk)[8], unsigned (&obw)[2], const bool direct) {
            const int m = m0 + wm * 128 + i * 32 + fr;
            // dropout: element index e = m * row_mult * N + n, pair index e >> 1 = P + c with P the pair of (this row, the wave's first
            // column + 4 fh) and c = j*16 + rg*4 (+1): x = lo(pair) * C1 ^ hi(pair) * C2 ^ key = (P_lo * C1 + c * C1) ^ (P_hi * C2 [+ C2 on carry]) ^ key
            uint32_t dP_lo = 0, dA1 = 0, dB1 = 0;
            if (DROP) {
                const unsigned long long P = ((unsigned long long)m * (unsigned long long)(p.drop.row_mult * (long long)p.N) + (unsigned)(n0 + wn * 64 + 4 * fh)) >> 1;
                dP_lo = (uint32_t)P;
                dA1 = dP_lo * 0x9E3779B1u;
                dB1 = (uint32_t)(P >> 32) * 0x85EBCA77u;
            }
            obw[0] = obw[1] = 0u;       // sign bits of this lane's 2 x 16 outputs, at their column positions (before the final << 4 fh)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int gi = j * 4 + rg;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][rg * 4 + e] * p.alpha + bias4[j][rg][e];
                        if (ACT == ACT_RELU && !PACKED_RELU) v[e] = fmaxf(v[e], 0.f);
                        else if (ACT == ACT_GELU) v[e] = gelu_f(v[e]);
                    }
                    if (DROP) {
                        unsigned keep = 0;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const uint32_t c = (uint32_t)(j * 16 + rg * 4 + h);
                            const uint32_t hb = (dP_lo + c < c) ? dB1 + 0x85EBCA77u : dB1;
                            const unsigned x = drop_mix((dA1 + c * 0x9E3779B1u) ^ hb ^ p.drop.key);
                            keep |= ((x & 0xffffu) >= p.drop.thr ? 1u : 0u) << (2 * h);
                            keep |= ((x >> 16) >= p.drop.thr ? 2u : 0u) << (2 * h);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * p.drop.scale : 0.f;
                    }
                    if (HAS_MASK) {
                        const u32x2 mk = HAS_RES ? *(const u32x2*)(p.relu_mask + (size_t)min(m, p.M - 1) * p.ldm + n0 + wn * 64 + j * 32 + 8 * rg + 4 * fh)
                                                 : *(const u32x2*)(Es + (e_wr ^ (gi << 4)));
                        if (!(bf_lo(mk[0]) > 0.f)) v[0] = 0.f;
                        if (!(bf_hi(mk[0]) > 0.f)) v[1] = 0.f;
                        if (!(bf_lo(mk[1]) > 0.f)) v[2] = 0.f;
                        if (!(bf_hi(mk[1]) > 0.f)) v[3] = 0.f;
                    }
                    if (HAS_BITS) {
                        const unsigned nib = mbits[i & 1][j] >> (8 * rg + 4 * fh);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (!((nib >> e) & 1u)) v[e] = 0.f;
                    }
                    if (HAS_RES) {
                        const u32x2 rs = *(const u32x2*)(Es + (e_wr ^ (gi << 4)));
                        v[0] += bf_lo(rs[0]); v[1] += bf_hi(rs[0]); v[2] += bf_lo(rs[1]); v[3] += bf_hi(rs[1]);
                    }
                    pk[gi] = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    if (PACKED_RELU) { pk[gi][0] = pk_max_i16(pk[gi][0], 0u); pk[gi][1] = pk_max_i16(pk[gi][1], 0u); }      // negative halves (and -0) -> +0
                    if (direct) *(u32x2*)(Es + (e_wr ^ (gi << 4))) = pk[gi];      // unpipelined flavours: straight into the staging buffer (no 16 live registers)
                    if (PACKED_RELU && p.bits_out) {   // outputs are >= +0: min(half, 1) per 16-bit half, bit 16 folded down to bit 1
                        const unsigned t0 = pk_min_i16(pk[gi][0], 0x00010001u), t1 = pk_min_i16(pk[gi][1], 0x00010001u);
                        obw[j] |= (((t0 | (t0 >> 15)) & 3u) | (((t1 | (t1 >> 15)) & 3u) << 2)) << (8 * rg);
                    }
                }
            }
        };
        auto write_slab = [&](u32x2 (&pk)[8]) {
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) *(u32x2*)(Es + (e_wr ^ (gi << 4))) = pk[gi];
        };
        auto read_slab = [&](u32x4 (&w)[4]) {
#pragma unroll
            for (int it = 0; it < 4; ++it) w[it] = *(const u32x4*)(Es + it_off(it) + e_rd);
        };
        auto store_slab = [&](const int i, u32x4 (&w)[4], unsigned (&obw)[2]) {
            const int mrow0 = m0 + wm * 128 + i * 32;      // wave-uniform
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r0 = mrow0 + it * 8;
                char* cb = C + ((size_t)r0 * p.ldc + n0 + wn * 64) * 2;
                if (p.dbg & 1) { asm volatile("" ::"v"(w[it])); }
                else if (r0 + 8 <= p.M) *(u32x4*)(cb + c_loff) = w[it];
                else if (r0 + (lo >> 3) < p.M) *(u32x4*)(cb + c_loff) = w[it];
            }
            if (PACKED_RELU && p.bits_out) {
                // lanes fr and fr+32 hold the two interleaved nibble sets of row fr: merge, then ONE 8-byte store per row (store
                // instructions, not bytes, are what the CU's store path charges for)
                const unsigned o0 = obw[0] << (4 * fh), o1 = obw[1] << (4 * fh);
                const auto s0 = __builtin_amdgcn_permlane32_swap(o0, o0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(o1, o1, false, false);
                const u32x2 ob = {o0 | s0[1], o1 | s1[1]};
                const int m = mrow0 + fr;
                if (fh == 0 && m < p.M) *(u32x2*)(p.bits_out + relu_bits_word(m, n0 + wn * 64, p.N)) = ob;
            }
        };
        if (!LDS_AUX) {
            u32x2 pkA[8], pkB[8];
            unsigned obA[2], obB[2];
            u32x4 w[4];
            if (HAS_BITS) load_bits(1, mbits[1], m0, n0, lo);
            convert(0, pkA, obA, false);
            write_slab(pkA);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("" ::: "memory");
                read_slab(w);
                asm volatile("" ::: "memory");
                if (i + 1 < 4) {      // under the read-back's latency
                    if (HAS_BITS && i + 2 < 4) load_bits(i + 2, mbits[i & 1], m0, n0, lo);
                    if (i & 1) convert(i + 1, pkA, obA, false); else convert(i + 1, pkB, obB, false);
                }
                if (i & 1) store_slab(i, w, obB); else store_slab(i, w, obA);
                asm volatile("" ::: "memory");
                if (i + 1 < 4) { if (i & 1) write_slab(pkA); else write_slab(pkB); }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (LDS_AUX) {
                    if (i + 2 < 4) load_aux(i + 2, auxrm[(i + 2) % 3], m0, n0, lo);
#pragma unroll
                    for (int it = 0; it < 4; ++it) *(u32x4*)(Es + it_off(it) + e_rd) = auxrm[i % 3][it];
                    __builtin_amdgcn_wave_barrier();
                }
                if (HAS_BITS && i + 1 < 4) load_bits(i + 1, mbits[(i + 1) & 1], m0, n0, lo);
                u32x2 pk[8];
                unsigned obw[2];
                u32x4 w[4];
                convert(i, pk, obw, true);
                __builtin_amdgcn_wave_barrier();
                read_slab(w);
                store_slab(i, w, obw);
                __builtin_amdgcn_wave_barrier();
            }
        }
        return (m0 + 256 <= p.M) && !(p.dbg & 3);
    }
};

// ACT / AUX (bit 0: +residual, bit 1: ReLU mask) are compile-time: a runtime-selected epilogue unrolled over the 32 accumulator
// pieces is ~100 KiB of code (128 inlined erff bodies ...) that evicts the main loop from the instruction cache once per tile.
template <int ACT, int AUX, bool DROP>
__device__ __forceinline__ void gemm_nt256k64_bf16_kernel_body(GemmNtArgs p) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* As = (bf16_t*)smem;                    // [NS64][256][64]
    bf16_t* Bs = As + NS64 * 256 * BK64;           // [NS64][256][64]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int ntn = (p.N + 255) / 256, MT = (p.M + 255) / 256;     // N % 128 == 0: the last n-tile may be a half tile (ViT-S widths 384 / 1152)
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3, lstride = gridDim.x >> 3;
    const int n_local = ((MT - xcd + 7) / 8) * ntn;
    if (lw >= n_local) return;
    const int my_tiles = (n_local - lw + lstride - 1) / lstride;
    const int nk = p.K / BK64;
    const int total = my_tiles * nk;

    int ldsoff[4], srow[4], schunk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rbase = (wid * 4 + j) * 8;
        srow[j] = rbase + (lane >> 3);
        schunk[j] = ((lane & 7) ^ ((srow[j] >> 1) & 7)) * 8;
        ldsoff[j] = __builtin_amdgcn_readfirstlane(rbase * BK64);
    }
    auto tile_origin = [&](int ti, int& m0, int& n0) {
        const int li = lw + ti * lstride;
        m0 = ((li / ntn) * 8 + xcd) * 256;
        n0 = (li % ntn) * 256;
    };
    uint32_t offB[4], offA[4];
    const char* a_base = nullptr;
    const char* b_base = nullptr;
    auto set_dma_tile = [&](int ti) {
        int m0d, n0d;
        tile_origin(ti, m0d, n0d);
        m0d = __builtin_amdgcn_readfirstlane(m0d);
        n0d = __builtin_amdgcn_readfirstlane(n0d);
        a_base = (const char*)p.A + (size_t)m0d * p.lda * 2;
        b_base = (const char*)p.B + (size_t)n0d * p.ldb * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            offA[j] = (uint32_t)(((min(m0d + srow[j], p.M - 1) - m0d) * p.lda + schunk[j]) * 2);
            offB[j] = (uint32_t)(((min(n0d + srow[j], p.N - 1) - n0d) * p.ldb + schunk[j]) * 2);      // half last n-tile: clamp (never stored)
        }
    };
    int d = 0, d_kt = 0, d_tile = 0;
    set_dma_tile(0);
    auto issue_next = [&]() {
        if (d >= total) return;
        const int st = d % NS64;
        const uint32_t kb = (uint32_t)(d_kt * BK64 * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base + (offA[j] + kb)),
                                             (__attribute__((address_space(3))) void*)(As + st * 256 * BK64 + ldsoff[j]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_base + (offB[j] + kb)),
                                             (__attribute__((address_space(3))) void*)(Bs + st * 256 * BK64 + ldsoff[j]), 16, 0, 0);
        }
        ++d;
        if (++d_kt == nk) {
            d_kt = 0;
            if (++d_tile < my_tiles) set_dma_tile(d_tile);
        }
    };

    f32x16 acc[4][2];
    const int fr = lane & 31, fh = lane >> 5;
    Nt256Epi<ACT, AUX, 2048, false, DROP> epi(p, smem + (size_t)NS64 * 512 * BK64 * 2 + (size_t)__builtin_amdgcn_readfirstlane(wid) * 4096, nullptr, wid, lane);
    for (int s = 0; s < NS64 - 1; ++s) issue_next();
    int g = 0;
    bool prev_full = true;
    for (int ti = 0; ti < my_tiles; ++ti) {
        int m0, n0;
        tile_origin(ti, m0, n0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt, ++g) {
            // stage g is the only DMA group in flight (NS64 = 2), except for the previous tile's 16 epilogue stores issued
            // after it: those may keep draining (VM counter retires in order)
            if (NS64 == 2) {
                if (ti > 0 && kt == 0 && prev_full) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt == nk - 1) epi.prefetch0(m0, n0);
            issue_next();
            const int st = g % NS64;
            const bf16_t* Ab = As + st * 256 * BK64;
            const bf16_t* Bb = Bs + st * 256 * BK64;
            // Fragment reads are software-pipelined by hand: the reads of k-step kk+1 are issued BEFORE the MFMAs of k-step kk (two
            // register sets), so an LDS latency is exposed once per K-tile instead of before every group of 4 MFMAs (what the compiler's
            // own schedule did: ds_read x4 -> s_waitcnt lgkmcnt(0) -> 4 MFMA -> ...).
            bf16x8 fa[2][4], fb[2][2];
            auto load_frags = [&](int kk, bf16x8 (&A4)[4], bf16x8 (&B2)[2]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r_ = wm * 128 + t * 32 + fr;
                    A4[t] = *(const bf16x8*)(Ab + r_ * BK64 + swz64(r_, kk * 2 + fh) * 8);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r_ = wn * 64 + t * 32 + fr;
                    B2[t] = *(const bf16x8*)(Bb + r_ * BK64 + swz64(r_, kk * 2 + fh) * 8);
                }
            };
            if (p.dbg & 64) continue;          // timing-only: DMA stream + barriers alone (what does the operand fetch path sustain?)
            load_frags(0, fa[0], fb[0]);
#pragma unroll
            for (int kk = 0; kk < BK64 / 16; ++kk) {
                if (kk + 1 < BK64 / 16) load_frags(kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma32(fb[kk & 1][j], fa[kk & 1][i], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        prev_full = epi.run(acc, m0, n0);      // exactly 16 stores were issued behind this wave's in-flight DMA group
    }
}
template <int ACT, int AUX, bool DROP>
__global__ void __launch_bounds__(NT256_THREADS, 2) gemm_nt256k64_bf16_kernel(GemmNtArgs p) { gemm_nt256k64_bf16_kernel_body<ACT, AUX, DROP>(p); }

// =================================================================================================
// 8-phase ("ping-pong") 256x256x64 NT kernel.  Same tile, wave layout (2(M) x 4(N) waves, 128 x 64 per wave) and epilogue as
// gemm_nt256k64; what changes is the operand pipeline.
//   * LDS ring of EIGHT 16-KiB half-tiles instead of two 64-KiB stages.  A K-tile is four half-tiles, in DMA order
//       B0 = cols {wn*64 + 0..31}  (j = 0)      B1 = cols {wn*64 + 32..63}  (j = 1)
//       A0 = rows {wm*128 + 0..63} (i = 0,1)    A1 = rows {wm*128 + 64..127} (i = 2,3)
//     each [128 rows][64 k] bf16, 128-byte rows (full-line LDS-DMA), XOR-swizzled like the stages of the 2-buffer kernel.
//   * A K-tile is computed as four phases, phase i = the wave's accumulator row block i (32 rows x 64 columns, 8 MFMAs).  Both B
//     half-tiles are read whole (all 64 k) into registers in phase 0 and kept for the K-tile, A row block i is read in phase i: every
//     LDS byte is read once per wave, and a half-tile's ring slot is free again two phases after its last read -- the ring holds
//     almost only data in flight.  The DMA stream runs P8_L = 6 half-tiles (96 KiB) ahead of the reads, one half-tile (2 DMA
//     instructions per wave) issued per phase, continuously across K-tiles and output tiles, waited for with a counted vmcnt
//     (never 0 in steady state).
//   * Two wave groups (wm = 0 / 1: one wave of each on every SIMD) run one barrier apart: while one group issues its MFMAs, the
//     other issues its fragment reads and DMA.  Each phase = [reads + DMA issue] barrier [8 MFMAs] barrier.
// Synchronisation (ticks = intervals between workgroup barriers; group 0: load(P) in tick 2P, compute(P) in tick 2P+1; group 1 one
// tick later; half-tile q = 4*ktile + type).  RAW: every wave waits for ITS pieces of the half-tiles first read in phase P+1 at the
// end of load(P), i.e. before a barrier that precedes any read of them: at P = 3 (mod 4) the next K-tile's B0, B1, A0 (q <= P+3 of
// the q <= P+6 issued: 3 half-tiles = 6 instructions may stay in flight), at P = 1 (mod 4) A1 (q <= P+2: 8 instructions).  WAR:
// half-tile q+8 is issued in load(q+2), two full phases after the last read of q (which is in phase <= q; the reads of a phase are
// retired by the lgkmcnt wait in front of its MFMAs, one barrier before the next phase of the same group).
#define P8_RING_BYTES 131072
template <int ACT, int AUX, bool DROP>
__device__ __forceinline__ void gemm_nt8p_bf16_kernel_body(GemmNtArgs p) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 2, wn = wid & 3;
    const int ntn = (p.N + 255) / 256, MT = (p.M + 255) / 256;
    const int xcd = blockIdx.x & 7, lw = blockIdx.x >> 3, lstride = gridDim.x >> 3;
    const int n_local = ((MT - xcd + 7) / 8) * ntn;
    if (lw >= n_local) return;
    const int my_tiles = (n_local - lw + lstride - 1) / lstride;
    const int nk = p.K / BK64;
    // tile list of this workgroup: li = lw, lw + lstride, ...; li = mq * ntn + nr -> m-block (mq*8 + xcd), n-tile nr.  Walked without integer
    // division (a division per tile on each of the two cursors cost ~1.3 k cycles per tile: cycle counters, profiles/r03_nt_ktile_position_cycles.txt)
    const int step_q = lstride / ntn, step_r = lstride % ntn;
    struct Cursor { int mq, nr; };
    auto advance = [&](Cursor& c) {
        c.mq += step_q;
        c.nr += step_r;
        if (c.nr >= ntn) { c.nr -= ntn; ++c.mq; }
    };
    Cursor c_cur{lw / ntn, lw % ntn}, d_cur = c_cur;

    // ---- DMA side.  One wave instruction = 8 half-tile rows x 128 B; wave w fills half-tile rows w*16 + j*8 + (lane>>3), j = 0,1.
    uint32_t offB[2][2], offA[2][2];          // [half][j]: byte offset of this lane's 16-byte chunk from the tile's operand base
    const char* a_base = nullptr;
    const char* b_base = nullptr;
    bool d_offsets_full = false;      // offA / offB hold the (tile-independent) offsets of a full tile
    auto set_dma_tile = [&]() {
        const int m0d = __builtin_amdgcn_readfirstlane((d_cur.mq * 8 + xcd) * 256), n0d = __builtin_amdgcn_readfirstlane(d_cur.nr * 256);
        a_base = (const char*)p.A + (size_t)m0d * p.lda * 2;
        b_base = (const char*)p.B + (size_t)n0d * p.ldb * 2;
        const bool full = m0d + 256 <= p.M && n0d + 256 <= p.N;
        if (full && d_offsets_full) return;      // only the M tail / the half last n-tile clamp rows: everything else keeps its lane offsets
        d_offsets_full = full;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rho = wid * 16 + (lane >> 3) + 8 * j;
            const int csw = ((lane & 7) ^ ((rho >> 1) & 7)) * 8;       // swizzle on the SOURCE address (the DMA writes lane-linear)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int arow = (rho >> 6) * 128 + h * 64 + (rho & 63);
                const int bcol = (rho >> 5) * 64 + h * 32 + (rho & 31);
                offA[h][j] = (uint32_t)(((min(m0d + arow, p.M - 1) - m0d) * p.lda + csw) * 2);      // M tail: clamp (never stored)
                offB[h][j] = (uint32_t)(((min(n0d + bcol, p.N - 1) - n0d) * p.ldb + csw) * 2);      // half last n-tile: clamp
            }
        }
    };
    int d_tile = 0, d_kt = 0, d_par = 0;
    bool d_live = true;
    const int wrow_off = __builtin_amdgcn_readfirstlane(wid * 2048);
    set_dma_tile();
    // ty: 0 = B0, 1 = B1, 2 = A0, 3 = A1 (compile-time at every call site); ring slot = 4 * (K-tile parity) + ty
    auto issue = [&](const int ty) {
        if (d_live && !(p.dbg & 32)) {      // timing-only: dbg & 32 = no DMA at all, dbg & 8 = always K-tile 0 of the first tile (L2-resident operands)
            const char* base = (p.dbg & 8) ? (ty >= 2 ? (const char*)p.A : (const char*)p.B) : (ty >= 2 ? a_base : b_base);
            const uint32_t kb = (p.dbg & 8) ? 0u : (uint32_t)(d_kt * BK64 * 2);
            char* dst = smem + (d_par * 4 + ty) * 16384 + wrow_off;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t off = (ty == 0 ? offB[0][j] : ty == 1 ? offB[1][j] : ty == 2 ? offA[0][j] : offA[1][j]) + kb;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
            }
        }
        if (ty == 3) {          // the K-tile's last half-tile: advance the DMA cursor
            d_par ^= 1;
            if (++d_kt == nk) {
                d_kt = 0;
                if (++d_tile < my_tiles) { advance(d_cur); set_dma_tile(); }
                else d_live = false;
            }
        }
    };

    // ---- fragment read addresses: A row (wm*64 + i'*32 + fr), B row (wn*32 + fr) of a half-tile, logical chunk kk*2 + fh
    uint32_t ra[4], rb[4];
    {
        const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ko = ((kk * 2 + fh) ^ ((fr >> 1) & 7)) * 16;
            ra[kk] = (uint32_t)(32768 + (wm * 64 + fr) * 128 + ko);
            rb[kk] = (uint32_t)((wn * 32 + fr) * 128 + ko);
        }
    }
    f32x16 acc[4][2];
    bf16x8 fa[4], fb0[4], fb1[4];
    // Epilogue staging lives in the ring: when a wave enters the epilogue, the DMA stream has issued the next tile's half-tiles
    // 0..5, so the two slots of types A0 / A1 of the K-tile after next (parity c_par ^ 1) hold no live data; wave w uses ITS OWN
    // 2-KiB stripes of them (rows w*16 .. w*16+15 -- the rows only wave w ever fills by DMA, so no other wave touches them, and its
    // own next DMA into them is issued after its last staging access has retired).  The 32 KiB behind the ring hold bias[0..N).
    float* bias_lds = (float*)(smem + P8_RING_BYTES);
    for (int i = tid; i < p.N; i += NT256_THREADS) bias_lds[i] = p.bias ? p.bias[i] : 0.f;
    Nt256Epi<ACT, AUX, 16384, true, DROP> epi(p, smem, bias_lds, wid, lane);

    // prologue: half-tiles 0..5 in flight, 0..2 (the first K-tile's B0, B1, A0) landed and published
    issue(0); issue(1); issue(2); issue(3); issue(0); issue(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind
    int c_par = 0;
    bool eb = false;       // the previous tile's epilogue left exactly 16 stores behind the DMA that was already in flight
#define P8_LDS(off) (*(const bf16x8*)(smem + (off)))
    // The MFMAs are pure register operations: nothing but data dependences keeps them between the two barriers of their phase.  Both
    // barriers are therefore inline asm that "modifies" the phase's accumulators (and clobbers memory, which pins the LDS reads and
    // the DMA issue on their side of it).
#define P8_COMPUTE(i_)                                                                                                   \
    asm volatile("s_barrier\n\ts_setprio 1" : "+v"(acc[i_][0]), "+v"(acc[i_][1])::"memory");                             \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                                   \
        acc[i_][0] = mfma32(fb0[kk], fa[kk], acc[i_][0]);                                                                \
        acc[i_][1] = mfma32(fb1[kk], fa[kk], acc[i_][1]);                                                                \
    }                                                                                                                    \
    asm volatile("s_setprio 0\n\ts_barrier" : "+v"(acc[i_][0]), "+v"(acc[i_][1])::"memory");
    // first K-tile of an output tile: the first MFMA of each accumulator takes a zero C operand (an inline constant) instead of 128
    // v_mov per wave clearing the accumulators (~1 k cycles per tile with two waves per SIMD competing for the VALU)
#define P8_COMPUTE_FIRST(i_)                                                                                             \
    asm volatile("s_barrier\n\ts_setprio 1" ::: "memory");                                                               \
    acc[i_][0] = mfma32(fb0[0], fa[0], zero16);                                                                          \
    acc[i_][1] = mfma32(fb1[0], fa[0], zero16);                                                                          \
    _Pragma("unroll") for (int kk = 1; kk < 4; ++kk) {                                                                   \
        acc[i_][0] = mfma32(fb0[kk], fa[kk], acc[i_][0]);                                                                \
        acc[i_][1] = mfma32(fb1[kk], fa[kk], acc[i_][1]);                                                                \
    }                                                                                                                    \
    asm volatile("s_setprio 0\n\ts_barrier" : "+v"(acc[i_][0]), "+v"(acc[i_][1])::"memory");
    // one K-tile = four phases; CMP = P8_COMPUTE / P8_COMPUTE_FIRST
#define P8_KTILE(CMP, KT)                                                                                                \
    {                                                                                                                    \
        const uint32_t kbase = (uint32_t)c_par * 65536u;                                                                 \
        c_par ^= 1;                                                                                                      \
        if (p.dbg & 4096) t_kt = __builtin_readcyclecounter();                                                           \
        /* ---- phase 0: both B half-tiles + A row block 0 */                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fb0[kk] = P8_LDS(kbase + rb[kk]);                               \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fb1[kk] = P8_LDS(kbase + 16384 + rb[kk]);                       \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fa[kk] = P8_LDS(kbase + ra[kk]);                                \
        issue(2);                                                                                                        \
        CMP(0)                                                                                                           \
        /* ---- phase 1 */                                                                                               \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fa[kk] = P8_LDS(kbase + 4096 + ra[kk]);                         \
        issue(3);                                                                                                        \
        if (!d_live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* tail of this workgroup's stream: nothing is issued any more */ \
        else if (eb) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      /* 8 + the epilogue's 16 stores (VM ops retire in order) */ \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                            \
        eb = false;                                                                                                      \
        CMP(1)                                                                                                           \
        /* ---- phase 2 */                                                                                               \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fa[kk] = P8_LDS(kbase + 16384 + ra[kk]);                        \
        issue(0);                                                                                                        \
        CMP(2)                                                                                                           \
        /* ---- phase 3 */                                                                                               \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) fa[kk] = P8_LDS(kbase + 16384 + 4096 + ra[kk]);                 \
        issue(1);                                                                                                        \
        if (!d_live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                            \
        if ((KT) == nk - 1) epi.prefetch0(m0, n0);      /* behind the counted wait: ordinary loads in front of it would only tighten it */ \
        CMP(3)                                                                                                           \
        if ((p.dbg & 4096) && ti > 0) {                                                                                  \
            _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                                \
                if (q == ((KT) < 8 ? (KT) : 7)) cyc_kt[q] += __builtin_readcyclecounter() - t_kt;                        \
        }                                                                                                                \
    }

    long long cyc_main = 0, cyc_epi = 0, t_mark = (p.dbg & 16) ? __builtin_readcyclecounter() : 0;      // timing-only instrumentation (dbg & 16)
    long long cyc_kt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_kt = 0;      // dbg & 4096: cycles by K-tile position inside a tile (first 8)
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int m0 = (c_cur.mq * 8 + xcd) * 256, n0 = c_cur.nr * 256;
        advance(c_cur);
        P8_KTILE(P8_COMPUTE_FIRST, 0)
#pragma clang loop unroll(disable)
        for (int kt = 1; kt < nk; ++kt) P8_KTILE(P8_COMPUTE, kt)
        if (ti == my_tiles - 1 && wm == 0) __builtin_amdgcn_s_barrier();      // pairs with group 1's last barrier
        if (p.dbg & 16) { const long long t = __builtin_readcyclecounter(); cyc_main += t - t_mark; t_mark = t; }
        epi.Es = smem + ((c_par ^ 1) * 4 + 2) * 16384 + wrow_off;
        eb = epi.run(acc, m0, n0);
        if (p.dbg & 16) { const long long t = __builtin_readcyclecounter(); cyc_epi += t - t_mark; t_mark = t; }
    }
    if ((p.dbg & 4096) && tid == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ((float*)p.C)[8192 + blockIdx.x * 8 + q] = (float)cyc_kt[q] / (float)(my_tiles > 1 ? my_tiles - 1 : 1);
    }
    if ((p.dbg & 16) && tid == 0) {      // cycles per K-tile of the main loop, cycles per tile of the epilogue, into the first floats of C
        ((float*)p.C)[blockIdx.x * 2] = (float)cyc_main / (float)(my_tiles * nk);
        ((float*)p.C)[blockIdx.x * 2 + 1] = (float)cyc_epi / (float)my_tiles;
    }
#undef P8_KTILE
#undef P8_COMPUTE_FIRST
#undef P8_LDS
#undef P8_COMPUTE
}
template <int ACT, int AUX, bool DROP>
__global__ void __launch_bounds__(NT256_THREADS, 2) gemm_nt8p_bf16_kernel(GemmNtArgs p) { gemm_nt8p_bf16_kernel_body<ACT, AUX, DROP>(p); }

template <int ACT, int AUX, bool DROP>
static int launch_nt8p_inst(const GemmNtArgs& p, int grid, hipStream_t stream) {
    const size_t lds = (size_t)P8_RING_BYTES + 32768;   // 160 KiB: ring + bias table (N <= 8192)
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_nt8p_bf16_kernel<ACT, AUX, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    // tower-groupable flavours: plain / ReLU x bias / residual (the fusion encoder's and the compressor's forwards at an acting step)
    if constexpr (ACT <= 1 && AUX <= 1) SVLA_LAUNCH((gemm_nt8p_bf16_kernel<ACT, AUX, DROP>), (gemm_nt8p_bf16_kernel_body<ACT, AUX, DROP>), NT256_THREADS, 2, dim3(grid), dim3(NT256_THREADS), lds, stream, p);
    else hipLaunchKernelGGL((gemm_nt8p_bf16_kernel<ACT, AUX, DROP>), dim3(grid), dim3(NT256_THREADS), lds, stream, p);
    return svla_launch_status();
}

template <int ACT, int AUX, bool DROP>
static int launch_nt256_inst(const GemmNtArgs& p, int grid, hipStream_t stream) {
    const size_t lds = (size_t)NS64 * 512 * BK64 * 2 + 32768;   // 160 KiB
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_nt256k64_bf16_kernel<ACT, AUX, DROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((gemm_nt256k64_bf16_kernel<ACT, AUX, DROP>), dim3(grid), dim3(NT256_THREADS), lds, stream, p);
    return svla_launch_status();
}
// Kernel choice (measured, tools/ab_gemm.py, profiles/r03_nt_ab.txt): with the trimmed epilogue the 8-phase kernel wins on every shape of the
// update (+2 ... +7 %); the 2-buffer kernel stays for N > 8192 (the LDS bias table) and as the A/B partner.  dbg bits 128 / 256 force the
// 2-buffer / the 8-phase kernel.
static inline bool nt_use_8p(const GemmNtArgs& p) {
    if (p.N > 8192 || (p.dbg & 128)) return false;
    return true;
}
// dropout is a compile-time flavour of the epilogue; instantiated for every (activation, operand) combination it can be requested with
#define NT256_CASE(A_, X_)                                                                                                               \
    return p.drop.thr ? (nt_use_8p(p) ? launch_nt8p_inst<A_, X_, true>(p, grid, stream) : launch_nt256_inst<A_, X_, true>(p, grid, stream))   \
                      : (nt_use_8p(p) ? launch_nt8p_inst<A_, X_, false>(p, grid, stream) : launch_nt256_inst<A_, X_, false>(p, grid, stream))
static int launch_nt256(const GemmNtArgs& p, int grid, hipStream_t stream) {
    const int aux = (p.residual ? 1 : 0) | (p.relu_mask ? 2 : 0);
    if (p.bits_in) {
        if (p.act != ACT_NONE) return SVLA_EINVAL;
        NT256_CASE(0, 4);
    }
    switch (p.act * 4 + aux) {
        case 0: NT256_CASE(0, 0);
        case 1: NT256_CASE(0, 1);
        case 2: NT256_CASE(0, 2);
        case 3: NT256_CASE(0, 3);
        case 4: NT256_CASE(1, 0);
        case 5: NT256_CASE(1, 1);
        case 6: NT256_CASE(1, 2);
        case 7: NT256_CASE(1, 3);
        case 8: NT256_CASE(2, 0);
        case 9: NT256_CASE(2, 1);
        case 10: NT256_CASE(2, 2);
        case 11: NT256_CASE(2, 3);
        default: return SVLA_EINVAL;
    }
}
#undef NT256_CASE

static int g_force_small_tile = 0, g_dbg = 0;
// SVLA_GEMM_LOG=<file>: one line "kernel M N K" per big-GEMM launch, in launch order -- tools/prof_summarize.py joins it with the rocprofv3
// counter rows of the same kernel names (i-th dispatch <-> i-th line), so HBM traffic is reported per (kernel, shape) and not as a launch-weighted mean
static char g_last_kernel[96] = "";
static int g_last_shape[3] = {0, 0, 0};
// extra = bytes per output row beyond the A and C rows (residual 2 N, sign bits N / 8): the profile summary counts them as algorithmic bytes;
// to_file = false: remember the name only (the 128-tile kernel also runs the M % 256 tails, which are not logged, so its dispatches cannot be joined by index)
static void gemm_log(const char* kernel, int M, int N, int K, int extra = 0, bool to_file = true) {
    static FILE* f = nullptr;
    static bool init = false;
    if (!init) {
        init = true;
        const char* path = getenv("SVLA_GEMM_LOG");
        if (path) f = fopen(path, "w");
    }
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s", kernel);
    g_last_shape[0] = M; g_last_shape[1] = N; g_last_shape[2] = K;
    static const bool log_all = getenv("SVLA_GEMM_LOG_ALL") != nullptr;      // shape studies (tools/): the 128-tile launches too -- such a log cannot be joined with counter rows
    if (f && (to_file || log_all)) { fprintf(f, "%s %d %d %d %d\n", kernel, M, N, K, extra); fflush(f); }
}
// name (and M, N, K as dispatched: full tiles only for the assembly kernels) of the kernel the last svla_gemm_nt_bf16 / svla_gemm_tn_f32acc call of this
// process launched for its main problem -- tests assert that an "assembly" test really ran the assembly kernel and not a silent fall-through
extern "C" int svla_gemm_last_kernel(char* name, int cap, int* mnk) {
    if (!name || cap <= 0) return SVLA_EINVAL;
    snprintf(name, (size_t)cap, "%s", g_last_kernel);
    if (mnk) { mnk[0] = g_last_shape[0]; mnk[1] = g_last_shape[1]; mnk[2] = g_last_shape[2]; }
    return SVLA_OK;
}
// one-time device facts / scratch of the assembly dispatchers.  Thread-safe (std::call_once); refuses to initialise lazily inside a stream capture
// (hipMalloc + a synchronous memset would invalidate it): call any GEMM once -- or svla_gemm_force_small_tile(0), which _lib.py does at load -- first.
struct GemmGlobals { int n_cu = 0; float* zero_bias = nullptr; int rc = 0; };
static GemmGlobals g_gg;
static std::once_flag g_gg_once;
static int gemm_globals_init() {
    std::call_once(g_gg_once, [] {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&g_gg.n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e == hipSuccess) e = hipMalloc(&g_gg.zero_bias, 4096 * sizeof(float));
        if (e == hipSuccess) e = hipMemset(g_gg.zero_bias, 0, 4096 * sizeof(float));
        g_gg.rc = (e == hipSuccess) ? svla_asm_preload() : (int)e;      // the assembly code object too: nothing is left to load inside a stream capture
    });
    return g_gg.rc;
}
static int gemm_globals(hipStream_t stream, const GemmGlobals** out) {
    if (!g_gg.n_cu) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return SVLA_EINVAL;
        const int rc = gemm_globals_init();
        if (rc) return rc;
    }
    *out = &g_gg;
    return 0;
}
// on = 0/1/2: normal dispatch / force the 128x128 kernels / force the 256-tile kernels wherever their shape constraints hold (tests); on = 10 + f: flags f of the 256-tile kernels -- timing-only ablations 1 / 2 / 64,
// 128 / 256 = force the 2-buffer kernel (gemm_nt256k64) / the 8-phase kernel (gemm_nt8p) (A/B comparisons)
extern "C" int svla_gemm_force_small_tile(int on) {
    if (gemm_globals_init()) return SVLA_EINVAL;
    if (on >= 10) { g_dbg = on - 10; g_force_small_tile = 0; }
    else { g_dbg = 0; g_force_small_tile = on; }
    return SVLA_OK;
}

// ---- A-stationary assembly kernels (asmgen/nt_as_gen.py): K = 512, bf16 output, full 256-row panels; the M % 256 tail rows run as a
// sub-problem on the 128-tile kernel (row0 keeps its dropout counters / sign-bit blocks on the global row index).
#define NT_AS_NOT_TAKEN (-12345)
struct NtAsKarg {      // = asmgen/nt_as_gen.py KARG
    const void* A; long lda; const void* B; long ldb; const float* bias; const void* res; int nr, flags; void* C; long ldc;
    int cmask, N; float alpha; int npanels; const void* bits; unsigned key, thr; float scale; int row_mult; const unsigned* seed_dev; unsigned stream_key; int grid;
};
static_assert(sizeof(NtAsKarg) == 128, "kernarg layout of the nt_as kernels");
// floor of the mid-M launch in 256-row panels (SVLA_NT_AS_MIN_PANELS: sweeps of tools/ab_midm.py; the cost model below decides above it)
static int nt_as_min_panels() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("SVLA_NT_AS_MIN_PANELS"); v = e ? atoi(e) : 8; }
    return v;
}
static int nt_as_try(const GemmNtArgs& p, hipStream_t stream) {
    if ((p.K != 512 && p.K != 384) || p.out_f32 || (p.N % 128) || p.N > 4096 || p.N < 128 || (p.dbg & 8192) || g_force_small_tile == 1) return NT_AS_NOT_TAKEN;
    if (p.relu_mask) return NT_AS_NOT_TAKEN;
    const int npanels = p.M / 256;
    // >= 2 panels per CU: the row-streaming launch (one persistent workgroup per CU, phases); fewer: the mid-M launch below
    if (npanels < nt_as_min_panels() && g_force_small_tile != 2) return NT_AS_NOT_TAKEN;
    if (npanels < 1) return NT_AS_NOT_TAKEN;
    const char* name = nullptr;
    if (p.residual) return NT_AS_NOT_TAKEN;
    if (p.bits_in) {      // input gradient under the ReLU sign bits (+ dropout scale as alpha): no bias
        if (p.act == ACT_NONE && !p.bias && !p.drop.thr && !p.bits_out) name = "svla_nt_as_f3";
    } else if (p.act == ACT_RELU && p.bits_out && p.alpha == 1.f) {
        // the dropout counter of the assembly kernel is 32 bits wide: element pairs of the whole logical tensor must fit
        if (!p.drop.thr) name = "svla_nt_as_f1";
        else if ((unsigned long long)p.M * (unsigned long long)p.drop.row_mult * (unsigned long long)p.N / 2 < 0xffffffffull) name = "svla_nt_as_f1d";
    } else if (p.act == ACT_NONE && !p.bits_out && !p.drop.thr && p.alpha == 1.f) name = "svla_nt_as_f0";
    if (p.K == 384) {      // the ViT-S / DINOv2-feature width: bias (qkv), bias + erf-GELU (fc1), bias + ReLU + sign bits (the policy's visual compressor)
        if (name && !strcmp(name, "svla_nt_as_f0")) name = "svla_nt_as_k384_f0";
        else if (name && !strcmp(name, "svla_nt_as_f1")) name = "svla_nt_as_k384_f1";
        else if (!p.bits_in && p.act == ACT_GELU && !p.bits_out && !p.drop.thr && p.alpha == 1.f) name = "svla_nt_as_k384_f2";
        else name = nullptr;
    }
#ifdef SVLA_ASM_DEBUG      // instrumented builds of tools/ only (SVLA_EXTRA_FLAGS=-DSVLA_ASM_DEBUG): svla_nt_as_f0_<variant>
    static char dbg_name[96];
    if (name && getenv("SVLA_NT_AS_VARIANT")) {
        snprintf(dbg_name, sizeof(dbg_name), "%s_%s", name, getenv("SVLA_NT_AS_VARIANT"));
        name = dbg_name;
    }
#endif
    if (!name || !svla_asm_has(name)) return NT_AS_NOT_TAKEN;
    // the sign-bit offset of a panel (panel * N * 32 bytes) is 32-bit scalar arithmetic in the kernel (asmgen/nt_as_gen.py bits_srd)
    if ((p.bits_in || p.bits_out) && (unsigned long long)p.M * (unsigned long long)p.N / 8 >= 0xffffffffull) return NT_AS_NOT_TAKEN;
    const GemmGlobals* gg = nullptr;
    { const int rc = gemm_globals(stream, &gg); if (rc) return rc; }
    const int n_cu = gg->n_cu;
    float* const zero_bias = gg->zero_bias;
    NtAsKarg k;
    memset(&k, 0, sizeof(k));
    k.A = p.A; k.lda = p.lda; k.B = p.B; k.ldb = p.ldb; k.bias = p.bias ? p.bias : zero_bias; k.C = p.C; k.ldc = p.ldc;
    {      // phase spread over workgroups: (workgroup & cmask) < NS/4 extra steps, cmask + 1 = the largest power of two <= NS/4
        int q = p.N / 256, m = 1;
        while (m * 2 <= q) m *= 2;
        k.cmask = q >= 1 ? m - 1 : 0;
    }
    k.N = p.N; k.alpha = p.alpha; k.npanels = npanels; k.bits = p.bits_in ? (const void*)p.bits_in : (const void*)p.bits_out;
    k.key = p.drop.key; k.thr = p.drop.thr; k.scale = p.drop.scale; k.row_mult = p.drop.row_mult; k.seed_dev = p.drop.seed_dev; k.stream_key = p.drop.stream_key;
    k.nr = p.N; k.flags = 0;
    k.grid = npanels < n_cu ? npanels : n_cu;
    int grid_y = 1;
    if (npanels < 2 * n_cu) {
        // mid-M (an acting step's 45 panels, the 233 of the batch-256 probe, the ViT's 216): too few panels to give every CU two sweeps.  The grid becomes
        // (panel slots) x (n-ranges): workgroup (x, y) sweeps columns [y nr, (y + 1) nr) of panels x, x + slots, ...; no phases (a workgroup that holds one
        // panel has nothing to de-phase).  nsplit minimises rounds * (steps per sweep + ~2.5 steps of prologue / drain) over the divisors of N / 128.
        // Cost model, calibrated on profiles/r05_midm_sweep.txt (asm side within ~10 % of the measurements): one n-step (64 columns of a 256-row panel) takes
        // ~2.76 us * K/512; a workgroup pays ~5 steps per panel it loads (the fragment-shaped A gather runs at ~13 B/clk) + its sweep; the tile kernels take
        // ~(6.2 + 11.6 K/512) us per round of 256 tiles.  The assembly launch is taken when it is at least 5 % ahead (a ragged M adds the tail launch).
        const int ns = p.N / 64;
        double best = 1e30;
        int best_d = 1, best_slots = k.grid;
        for (int d = 1; d <= p.N / 128; ++d) {
            if ((p.N / 128) % d) continue;
            int slots = npanels * d <= n_cu ? npanels : n_cu / d;
            if (slots < 1) break;
            const int rounds = (npanels + slots - 1) / slots;
            const double cost = rounds * ((double)ns / d + 5.0);
            if (cost < best - 1e-9) { best = cost; best_d = d; best_slots = slots; }
        }
        if (g_force_small_tile != 2) {
            const long tiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
            const double hip = (double)((tiles + n_cu - 1) / n_cu) * (6.2 + 11.6 * p.K / 512.0);
            const double asm_cost = best * 2.76 * p.K / 512.0 + ((p.M % 256) ? 15.0 : 0.0);      // a ragged M: + the tail launch behind it (~15 us in a stream, r05_vit_kernel_stats.txt)
            if (asm_cost > 0.95 * hip) return NT_AS_NOT_TAKEN;
        }
        grid_y = best_d;
        k.grid = best_slots;
        k.nr = p.N / best_d;
        k.flags = 1;
        k.cmask = 0;
    }
#ifdef SVLA_ASM_DEBUG
    if (getenv("SVLA_NT_AS_DBGBUF")) k.bits = (const void*)strtoull(getenv("SVLA_NT_AS_DBGBUF"), nullptr, 16);      // timing builds of tools/time_nt_as.py
#endif
    gemm_log(name, npanels * 256, p.N, p.K, (p.bits_in || p.bits_out) ? p.N / 8 : 0);
    const int rc = svla_asm_launch2(name, &k, sizeof(k), k.grid, grid_y, 256, stream);
    if (rc) return rc;
    const int tail = p.M - npanels * 256;
    if (tail > 0) {
        GemmNtArgs q = p;
        const size_t r0 = (size_t)npanels * 256;
        q.A = p.A + r0 * p.lda;
        q.C = (void*)((bf16_t*)p.C + r0 * p.ldc);
        if (p.residual) q.residual = p.residual + r0 * p.ldr;
        if (p.bits_in) q.bits_in = p.bits_in + (r0 >> 5) * (size_t)(p.N >> 6) * 256;
        if (p.bits_out) q.bits_out = p.bits_out + (r0 >> 5) * (size_t)(p.N >> 6) * 256;
        q.M = tail;
        q.row0 = (int)r0;
        const int mt = (tail + BM - 1) / BM, nt = p.N / BN;
        const size_t lds = BM * (BN + 4) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        SVLA_LAUNCH(gemm_nt_bf16_kernel, gemm_nt_bf16_kernel_body, NTHREADS, 2, dim3(mt * nt), dim3(NTHREADS), lds, stream, q);
        return svla_launch_status();
    }
    return SVLA_OK;
}

// ---- output-stationary assembly kernels (asmgen/nt_os_gen.py): K > 512 (or K = 512 with a residual), bias / residual epilogues without dropout, N % 256 == 0,
// K % 128 == 0, full 256-row tiles; the M % 256 tail rows run on the 128-tile kernel.  g_dbg & 16384 = off (A/B: tools/ab_nt_os.py).
struct NtOsKarg {      // = asmgen/nt_os_gen.py KARG
    const void* A; long lda; const void* B; long ldb; const float* bias; const void* res; long ldr; void* C; long ldc;
    int M, N, K, ntn, ntiles, grid;
};
static_assert(sizeof(NtOsKarg) == 96, "kernarg layout of the nt_os kernels");
static int nt_os_try(const GemmNtArgs& p, hipStream_t stream) {
    if (p.out_f32 || (p.N % 256) || p.N > 4096 || (p.K % 128) || p.K < 384 || (p.dbg & (8192 | 16384)) || g_force_small_tile == 1) return NT_AS_NOT_TAKEN;
    if (p.relu_mask || p.bits_in || p.bits_out || p.drop.thr || p.act != ACT_NONE || p.alpha != 1.f) return NT_AS_NOT_TAKEN;
    const int mtiles = p.M / 256, ntn = p.N / 256;
    if ((long)mtiles * ntn < 1024 && g_force_small_tile != 2) return NT_AS_NOT_TAKEN;      // fewer than four tiles per CU: the two-workgroup-per-CU tile kernel balances better
    if (mtiles < 1) return NT_AS_NOT_TAKEN;
    const char* name = p.bias ? (p.residual ? "svla_nt_os_br" : "svla_nt_os_b") : (p.residual ? "svla_nt_os_r" : "svla_nt_os_p");
#ifdef SVLA_ASM_DEBUG      // timing-only builds of tools/var_nt_os.py: svla_nt_os_r_<variant>
    static char dbg_name[96];
    if (getenv("SVLA_NT_OS_VARIANT")) {
        snprintf(dbg_name, sizeof(dbg_name), "%s_%s", name, getenv("SVLA_NT_OS_VARIANT"));
        name = dbg_name;
    }
#endif
    if (!svla_asm_has(name)) return NT_AS_NOT_TAKEN;
    const GemmGlobals* gg = nullptr;
    { const int rc = gemm_globals(stream, &gg); if (rc) return rc; }
    const int n_cu = gg->n_cu;
    NtOsKarg k;
    memset(&k, 0, sizeof(k));
    k.A = p.A; k.lda = p.lda; k.B = p.B; k.ldb = p.ldb; k.bias = p.bias; k.res = p.residual; k.ldr = p.ldr; k.C = p.C; k.ldc = p.ldc;
    k.M = mtiles * 256; k.N = p.N; k.K = p.K; k.ntn = ntn; k.ntiles = mtiles * ntn; k.grid = k.ntiles < n_cu ? k.ntiles : n_cu;
    gemm_log(name, k.M, p.N, p.K, p.residual ? 2 * p.N : 0);
    const int rc = svla_asm_launch(name, &k, sizeof(k), k.grid, 256, stream);
    if (rc) return rc;
    const int tail = p.M - mtiles * 256;
    if (tail > 0) {
        GemmNtArgs q = p;
        const size_t r0 = (size_t)mtiles * 256;
        q.A = p.A + r0 * p.lda;
        q.C = (void*)((bf16_t*)p.C + r0 * p.ldc);
        if (p.residual) q.residual = p.residual + r0 * p.ldr;
        q.M = tail;
        q.row0 = (int)r0;
        const int mt = (tail + BM - 1) / BM, nt = p.N / BN;
        const size_t lds = BM * (BN + 4) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        SVLA_LAUNCH(gemm_nt_bf16_kernel, gemm_nt_bf16_kernel_body, NTHREADS, 2, dim3(mt * nt), dim3(NTHREADS), lds, stream, q);
        return svla_launch_status();
    }
    return SVLA_OK;
}

// fewest 256x256 tiles for which the 256-tile kernel is chosen over the 128-tile one (SVLA_NT256_MIN_TILES: sweep of tools/acting_force_probe.py)
static int nt256_min_tiles() {
    static int v = 0;
    if (!v) { const char* e = getenv("SVLA_NT256_MIN_TILES"); v = e ? atoi(e) : 160; }
    return v;
}

extern "C" int svla_gemm_nt_bf16(const bf16_t* A, long lda, const bf16_t* B, long ldb, const float* bias,
                                 const bf16_t* residual, long ldr, const bf16_t* relu_mask, long ldm, void* C, long ldc,
                                 int M, int N, int K, int act, int out_f32, float alpha, unsigned char* relu_bits_out,
                                 const unsigned char* relu_bits, const svla_dropout* drop, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (N % BN) || (K % BK)) return SVLA_EINVAL;
    if ((relu_bits && (relu_mask || residual || out_f32)) || (relu_bits_out && (act != ACT_RELU || out_f32 || residual || relu_mask || relu_bits))) return SVLA_EINVAL;
    if (act < ACT_NONE || act > ACT_GELU) return SVLA_EINVAL;
    if ((lda % 8) || (ldb % 8) || (ldc % (out_f32 ? 4 : 8)) || (residual && (ldr % 8)) || (relu_mask && (ldm % 8))) return SVLA_EINVAL;
    GemmNtArgs p{A, lda, B, ldb, bias, residual, ldr, relu_mask, ldm, C, ldc, M, N, K, act, out_f32, alpha, relu_bits_out, relu_bits, drop_cfg(drop), 0, g_dbg, 0.f};
    {
        const int rc = nt_as_try(p, (hipStream_t)stream);      // K = 512 row-streaming GEMMs: the A-stationary assembly kernels
        if (rc != NT_AS_NOT_TAKEN) return rc;
    }
    {
        const int rc = nt_os_try(p, (hipStream_t)stream);      // K > 512 without dropout: the output-stationary assembly kernels
        if (rc != NT_AS_NOT_TAKEN) return rc;
    }
    // N % 256 == 128 with N >= 384 (the ViT-S widths 384 and 1152): the last n-tile is a half tile (75 % / 90 % of the MFMA work useful) --
    // still well ahead of the 128-tile kernel
    if (!out_f32 && ((N % 256) == 0 || ((N % 128) == 0 && N >= 384)) && (K % BK64) == 0 && K >= 2 * BK64 && ((long)((M + 255) / 256) * ((N + 255) / 256) >= nt256_min_tiles() || g_force_small_tile == 2) && g_force_small_tile != 1) {
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            HIP_CHECK_RET(hipGetDevice(&dev));
            HIP_CHECK_RET(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
            n_cu = (n_cu / 8) * 8;
            if (n_cu < 8) n_cu = 8;
        }
        const int ntiles = ((M + 255) / 256) * ((N + 255) / 256);
        gemm_log(nt_use_8p(p) ? "gemm_nt8p_bf16_kernel" : "gemm_nt256k64_bf16_kernel", M, N, K, (p.residual ? 2 * N : 0) + ((p.bits_in || p.bits_out) ? N / 8 : 0) + (p.relu_mask ? 2 * N : 0));
        int grid = n_cu;                       // persistent: one 512-thread workgroup (160 KiB LDS) per CU
        while (grid > 8 && (grid / 8) * 8 > ntiles) grid -= 8;
        return launch_nt256(p, grid, (hipStream_t)stream);
    }
    gemm_log("gemm_nt_bf16_kernel", M, N, K, 0, false);
    const int mt = (M + BM - 1) / BM, nt = N / BN;
    const size_t lds = BM * (BN + 4) * sizeof(float);  // 66 KiB: max(NST operand stages 64 KiB, fp32 epilogue tile)
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    SVLA_LAUNCH(gemm_nt_bf16_kernel, gemm_nt_bf16_kernel_body, NTHREADS, 2, dim3(mt * nt), dim3(NTHREADS), lds, (hipStream_t)stream, p);
    return svla_launch_status();
}

// RMSNorm(A) . W^T for small M (the frozen T5 encoder's and the llama decoder's pre-norm linears in an acting step: M = 64 ... 768 rows, where the norm was a
// launch of its own in front of every such GEMM): 128-tile kernel, the row statistics fall out of the A fragments, gamma is folded into W by the caller.
extern "C" int svla_gemm_nt_rmsa_bf16(const bf16_t* A, long lda, const bf16_t* B, long ldb, const float* bias, const bf16_t* residual, long ldr,
                                      void* C, long ldc, int M, int N, int K, int act, float eps, const svla_dropout* drop, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (N % BN) || (K % BK) || !(eps > 0.f)) return SVLA_EINVAL;
    if (act < ACT_NONE || act > ACT_GELU) return SVLA_EINVAL;
    if ((lda % 8) || (ldb % 8) || (ldc % 8) || (residual && (ldr % 8))) return SVLA_EINVAL;
    GemmNtArgs p{A, lda, B, ldb, bias, residual, ldr, nullptr, 0, C, ldc, M, N, K, act, 0, 1.f, nullptr, nullptr, drop_cfg(drop), 0, 0, eps};
    gemm_log("gemm_nt_bf16_kernel", M, N, K, 0, false);
    const int mt = (M + BM - 1) / BM, nt = N / BN;
    const size_t lds = BM * (BN + 4) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    SVLA_LAUNCH(gemm_nt_bf16_kernel, gemm_nt_bf16_kernel_body, NTHREADS, 2, dim3(mt * nt), dim3(NTHREADS), lds, (hipStream_t)stream, p);
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
    DetCfg det;                   // deterministic mode: fixed-point shadow of dW (common.h)
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
                grad_add(p.det, p.dW + (size_t)n * p.ldw + k, acc[i][j][r]);
            }
}

extern "C" int svla_colsum_bf16(const bf16_t* dY, long ldy, int M, int N, int row_stride, float* db, void* stream);

// =================================================================================================
// 256x256 output-tile weight-gradient kernel: dW[256 n x 256 k] += dY[chunk rows, n]^T . X[chunk rows, k] per workgroup
// (8 waves = 2(n) x 4(k), wave tile 128 x 64), 32 reduction rows per stage, 4-stage LDS-DMA pipeline (same counted-vmcnt
// scheme as gemm_nt256).  LDS rows are 512 B; the XOR swizzle of the 16-byte chunk index by (row & 3) << 2 is applied to
// the DMA source address and to the ds_read_b64_tr_b16 gathers (conflict-free).  Optional fused bias gradient:
// db[n] += sum_m dY[m, n], accumulated from the dY fragments already in registers by the k-tile-0 / k-wave-0 waves.
#define TN256_ROWS 64
#define TN_NS 2       // 2 x (64 rows x 256 cols x 2 operands) = 128 KiB; one barrier per 32 MFMAs per wave
__device__ __forceinline__ bf16x8 frag_tr256(const bf16_t* tile, int step, int col0, int lane) {
    const int pl = lane & 15, q = lane >> 4;
    const int colq = col0 + 16 * (q & 1) + 4 * (pl & 3);
    const int r0 = step * 16 + 8 * (q >> 1) + (pl >> 2);
    const int r1 = r0 + 4;
    const bf16x4 lo = lds_tr16_b64(tile + r0 * 256 + (((colq >> 3) ^ ((r0 & 3) << 2)) << 3) + (colq & 7));
    const bf16x4 hi = lds_tr16_b64(tile + r1 * 256 + (((colq >> 3) ^ ((r1 & 3) << 2)) << 3) + (colq & 7));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

struct GemmTn256Args {
    const bf16_t* dY; long ldy;
    const bf16_t* X; long ldx;
    float* dW; long ldw;
    float* db;
    int M, N, K, chunk_rows;
    int dbg;      // timing-only ablations of gemm_tn8p: 64 = DMA stream + barriers alone, 32 = no DMA (fragment reads + MFMAs + barriers), 16 = nt loads
    DetCfg det;   // deterministic mode: fixed-point shadow of dW / db (common.h)
};

__global__ void __launch_bounds__(NT256_THREADS, 2) gemm_tn256_bf16_kernel(GemmTn256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ys = (bf16_t*)smem;                          // [TN_NS][64][256]
    bf16_t* Xs = Ys + TN_NS * TN256_ROWS * 256;          // [TN_NS][64][256]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid >> 2, wk = wid & 3;
    const int ntk = p.K / 256, ntile = (p.N / 256) * ntk;
    // XCD-aware order: the output tiles that re-read one chunk of rows run on the same XCD's L2 (measured without it:
    // FETCH_SIZE = 2x the algorithmic bytes, every (n-tile, k-tile) pair pulling its operands from HBM again)
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = vid % ntile, chunk = vid / ntile;
    const int n0 = (tile / ntk) * 256, k0 = (tile % ntk) * 256;
    const int mbeg = chunk * p.chunk_rows;
    const int mend = min(p.M, mbeg + p.chunk_rows);      // multiple of 32 (launcher)
    const int nst = (mend - mbeg) / TN256_ROWS;
    if (nst <= 0) return;

    // DMA: one wave instruction = 1 KiB = 2 tile rows (full 512-byte row segments); 32 instructions per operand per stage,
    // 4 dY + 4 X per wave
    const bf16_t* gy[4];
    const bf16_t* gx[4];
    int ldsoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rb = (wid * 4 + j) * 2;
        const int row = rb + (lane >> 5);
        const int c = (lane & 31) ^ ((row & 3) << 2);
        gy[j] = p.dY + (size_t)(mbeg + row) * p.ldy + n0 + c * 8;
        gx[j] = p.X + (size_t)(mbeg + row) * p.ldx + k0 + c * 8;
        ldsoff[j] = __builtin_amdgcn_readfirstlane(rb * 256);
    }
    auto stage = [&](int st, int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gy[j] + (size_t)t * TN256_ROWS * p.ldy),
                                             (__attribute__((address_space(3))) void*)(Ys + st * TN256_ROWS * 256 + ldsoff[j]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gx[j] + (size_t)t * TN256_ROWS * p.ldx),
                                             (__attribute__((address_space(3))) void*)(Xs + st * TN256_ROWS * 256 + ldsoff[j]), 16, 0, 0);
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // bias gradient (column sums of dY) rides along in the k-tile-0 workgroups as ONE extra MFMA per 16 rows per wave against a
    // fragment of ones: wave (wn, wk) covers the 32 columns of fragment u = wk of its 128-column half.  (Summing the
    // fragments on the VALU stalls the wave on the LDS reads ahead of the MFMAs: measured +14...27 % on the whole launch.)
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
    // ... and the 16-row steps are dealt round-robin to the ntk workgroups that stream the same dY columns, so no workgroup
    // falls behind its L2 partners
    const bool do_bias = p.db != nullptr;
    int bturn = (tile % ntk);          // this workgroup takes a step when bturn == 0

    stage(0, 0);
    for (int t = 0; t < nst; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage t is the only DMA group in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < nst) stage((t + 1) % TN_NS, t + 1);
        const bf16_t* Yb = Ys + (t % TN_NS) * TN256_ROWS * 256;
        const bf16_t* Xb = Xs + (t % TN_NS) * TN256_ROWS * 256;
#pragma unroll
        for (int s2 = 0; s2 < TN256_ROWS / 16; ++s2) {
            bf16x8 fy[4], fx[2];
#pragma unroll
            for (int u = 0; u < 4; ++u) fy[u] = frag_tr256(Yb, s2, wn * 128 + u * 32, lane);
#pragma unroll
            for (int u = 0; u < 2; ++u) fx[u] = frag_tr256(Xb, s2, wk * 64 + u * 32, lane);
            if (do_bias) {
                if (bturn == 0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (u == wk) accb = mfma32(fy[u], ones, accb);
                }
                bturn = (bturn + 1 == ntk) ? 0 : bturn + 1;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = mfma32(fy[i], fx[j], acc[i][j]);
        }
    }
    // acc[i][j][reg]: n = n0 + wn*128 + i*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5);  k = k0 + wk*64 + j*32 + (lane&31)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int k = k0 + wk * 64 + j * 32 + (lane & 31);
                grad_add(p.det, p.dW + (size_t)n * p.ldw + k, acc[i][j][r]);
            }
    if (do_bias && (lane & 31) == 0) {   // every column of accb holds the same sums: lanes 0 and 32 publish their 16 rows
#pragma unroll
        for (int r = 0; r < 16; ++r)
            grad_add(p.det, p.db + n0 + wn * 128 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), accb[r]);
    }
}

// =================================================================================================
// 8-phase ("ping-pong") variant of the 256x256 weight-gradient kernel: same output tile, wave layout (2(n) x 4(k), 128 x 64 per
// wave), fragment gathers and epilogue as gemm_tn256; the operand pipeline is the one of gemm_nt8p.  One reduction step (16 rows
// of dY and X: 16 x 512 B + 16 x 512 B = one 16-KiB ring slot, filled by 2 DMA instructions per wave) is one phase:
// [12 transposed fragment reads + DMA issue of step P+6] barrier [8 MFMAs] barrier, the two wave groups (n halves) one barrier
// apart.  A slot is read in exactly one phase, so with 8 slots the DMA stream runs 6 steps (96 KiB) ahead; every wave waits for
// its pieces of step P+1 at the end of load(P) (5 younger steps = 10 instructions may stay in flight).
#define TNP_L 6
__global__ void __launch_bounds__(NT256_THREADS, 2) gemm_tn8p_bf16_kernel(GemmTn256Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wn = wid >> 2, wk = __builtin_amdgcn_readfirstlane(wid & 3);
    const int ntk = p.K / 256, ntile = (p.N / 256) * ntk;
    const int vid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = vid % ntile, chunk = vid / ntile;
    const int n0 = (tile / ntk) * 256, k0 = (tile % ntk) * 256;
    const int mbeg = chunk * p.chunk_rows;
    const int mend = min(p.M, mbeg + p.chunk_rows);      // multiple of 64 (launcher)
    const int nph = (mend - mbeg) / 16;
    if (nph <= 0) return;

    // DMA: wave w fills rows 2w, 2w+1 of the step's dY part (slot + 0) and of its X part (slot + 8192)
    const int drow = wid * 2 + (lane >> 5);
    const int dchunk = (lane & 31) ^ ((drow & 3) << 2);
    const uint32_t offy = (uint32_t)((drow * p.ldy + n0 + dchunk * 8) * 2), offx = (uint32_t)((drow * p.ldx + k0 + dchunk * 8) * 2);
    const char* ybase = (const char*)p.dY + (size_t)mbeg * p.ldy * 2;
    const char* xbase = (const char*)p.X + (size_t)mbeg * p.ldx * 2;
    const size_t ystep = (size_t)16 * p.ldy * 2, xstep = (size_t)16 * p.ldx * 2;
    const int wrow_off = __builtin_amdgcn_readfirstlane(wid * 1024);
    int dq = 0;
    auto issue = [&]() {
        if (dq < nph && !(p.dbg & 32)) {
            char* dst = smem + (dq & 7) * 16384 + wrow_off;
            if (p.dbg & 16) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ybase + offy),
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 2);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xbase + offx),
                                                 (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 2);
            } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ybase + offy),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xbase + offx),
                                             (__attribute__((address_space(3))) void*)(dst + 8192), 16, 0, 0);
            }
            if (!(p.dbg & 8)) {      // timing-only (8): re-read the same 16 rows (operands served by the L2)
                ybase += ystep;
                xbase += xstep;
            }
        }
        ++dq;
    };
    // fragment gather addresses inside a slot (see frag_tr256; rows r0 and r0 + 4 of the 16-row step)
    uint32_t ay[4], ax[2];
    {
        const int pl = lane & 15, q = lane >> 4;
        const int r0 = 8 * (q >> 1) + (pl >> 2);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int colq = wn * 128 + u * 32 + 16 * (q & 1) + 4 * (pl & 3);
            ay[u] = (uint32_t)((r0 * 256 + (((colq >> 3) ^ ((r0 & 3) << 2)) << 3) + (colq & 7)) * 2);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int colq = wk * 64 + u * 32 + 16 * (q & 1) + 4 * (pl & 3);
            ax[u] = (uint32_t)(8192 + (r0 * 256 + (((colq >> 3) ^ ((r0 & 3) << 2)) << 3) + (colq & 7)) * 2);
        }
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};   // bf16 1.0
    const bool do_bias = p.db != nullptr;
    int bturn = (tile % ntk);          // this workgroup takes a 16-row step of the bias gradient when bturn == 0

#pragma unroll
    for (int s = 0; s < TNP_L; ++s) issue();
    // steps 0 and 1 landed (4 younger steps = 8 instructions may stay in flight) and published
    if (nph > TNP_L) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long long t_begin = (p.dbg & 4) ? __builtin_readcyclecounter() : 0;
    // Transposed fragment gathers from inline asm: behind the builtin, hipcc cannot tell the reads from the LDS-DMA writes still in
    // flight and drains the whole DMA queue (s_waitcnt vmcnt(0)) in front of the first one.  Rows r0 and r0 + 4 (+ 4 * 512 B).
    // ONE barrier per phase, two fragment register sets: phase P = [step P's fragments retire] [gather step P+1 (its slot was published
    // by the previous barrier)] [8 MFMAs of step P] [DMA issue of step P+6] [wait: step P+2 landed] barrier.  The two waves of a SIMD
    // interleave their MFMAs and gathers freely.  (The two-barrier ping-pong of the NT kernel, gathering step P in its own load segment,
    // ran at 837 cycles per phase without any DMA and 1 098 with it; MFMA-bound is 512.)
    bf16x4 ylo[2][4], yhi[2][4], xlo[2][2], xhi[2][2];
#define TNP_READS(SET_, SB_)                                                                                                                  \
    {                                                                                                                                         \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                                                         \
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(ylo[SET_][u]), "=&v"(yhi[SET_][u]) : "v"((SB_) + ay[u]) : "memory"); \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                                                                         \
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:2048" : "=&v"(xlo[SET_][u]), "=&v"(xhi[SET_][u]) : "v"((SB_) + ax[u]) : "memory"); \
    }
#define TNP_PHASE(CUR_, NXT_)                                                                                                                 \
    {                                                                                                                                         \
        const bool bt = do_bias && bturn == 0;                                                                                                \
        if (do_bias) bturn = (bturn + 1 == ntk) ? 0 : bturn + 1;                                                                              \
        /* this step's fragment reads (issued one phase ago) retire; the asm "modifies" fragments and accumulators, which pins the MFMAs */   \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                   \
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(accb), \
                       "+v"(ylo[CUR_][0]), "+v"(yhi[CUR_][0]), "+v"(ylo[CUR_][1]), "+v"(yhi[CUR_][1]), "+v"(ylo[CUR_][2]), "+v"(yhi[CUR_][2]), "+v"(ylo[CUR_][3]), "+v"(yhi[CUR_][3]), \
                       "+v"(xlo[CUR_][0]), "+v"(xhi[CUR_][0]), "+v"(xlo[CUR_][1]), "+v"(xhi[CUR_][1])::"memory");                               \
        if (ph + 1 < nph) TNP_READS(NXT_, (uint32_t)((ph + 1) & 7) * 16384u)                                                                  \
        bf16x8 fy[4], fx[2];                                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) fy[u] = bf16x8{ylo[CUR_][u][0], ylo[CUR_][u][1], ylo[CUR_][u][2], ylo[CUR_][u][3], yhi[CUR_][u][0], yhi[CUR_][u][1], yhi[CUR_][u][2], yhi[CUR_][u][3]}; \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) fx[u] = bf16x8{xlo[CUR_][u][0], xlo[CUR_][u][1], xlo[CUR_][u][2], xlo[CUR_][u][3], xhi[CUR_][u][0], xhi[CUR_][u][1], xhi[CUR_][u][2], xhi[CUR_][u][3]}; \
        if (bt) {      /* wave-uniform; this wave's 32 bias columns are fragment u = wk */                                                    \
            const bf16x8 fsel = wk == 0 ? fy[0] : wk == 1 ? fy[1] : wk == 2 ? fy[2] : fy[3];                                                  \
            accb = mfma32(fsel, ones, accb);                                                                                                  \
        }                                                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fy[i], fx[j], acc[i][j]);                                        \
        issue();                                                                                                                              \
        if (dq <= nph) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      /* steady state: steps ph+3 .. ph+6 may stay in flight */        \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                /* tail of the chunk */                                          \
        asm volatile("s_barrier"                                                                                                              \
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(accb)::"memory"); \
    }
    if (p.dbg & 64) {        // timing-only: DMA stream + barriers alone
        for (int ph = 0; ph < nph; ++ph) {
            issue();
            if (dq <= nph) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    } else {
        TNP_READS(0, 0u)         // step 0 (published by the barrier above)
#pragma clang loop unroll(disable)
        for (int ph = 0; ph < nph; ph += 2) {      // nph is a multiple of 4 (64-row chunks)
            TNP_PHASE(0, 1)
            ++ph;
            TNP_PHASE(1, 0)
            --ph;
        }
    }
#undef TNP_PHASE
#undef TNP_READS
    if (p.dbg & 4) {      // timing-only: shader cycles per phase of this workgroup instead of the gradient
        if (tid == 0) p.dW[blockIdx.x] = (float)(__builtin_readcyclecounter() - t_begin) / (float)nph;
        return;
    }
    // acc[i][j][reg]: n = n0 + wn*128 + i*32 + (reg&3) + 8*(reg>>2) + 4*(lane>>5);  k = k0 + wk*64 + j*32 + (lane&31)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int k = k0 + wk * 64 + j * 32 + (lane & 31);
                grad_add(p.det, p.dW + (size_t)n * p.ldw + k, acc[i][j][r]);
            }
    if (do_bias && (lane & 31) == 0) {   // every column of accb holds the same sums: lanes 0 and 32 publish their 16 rows
#pragma unroll
        for (int r = 0; r < 16; ++r)
            grad_add(p.det, p.db + n0 + wn * 128 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), accb[r]);
    }
}

extern "C" int svla_gemm_tn_f32acc(const bf16_t* dY, long ldy, const bf16_t* X, long ldx, float* dW, long ldw, float* db, int M,
                                   int N, int K, void* stream) {
    if (M <= 0 || (N % 128) || (K % 128) || (ldy % 8) || (ldx % 8)) return SVLA_EINVAL;
    if (g_force_small_tile == 2 && (N % 256) == 0 && (K % 256) == 0 && M >= 2 * TN256_ROWS && (M % TN256_ROWS) != 0) {
        // forced big-tile mode (tests: the reference goldens through the kernels that carry the update): whole 64-row groups on the 256-tile
        // kernel, the ragged tail on the small one
        const int mb = M / TN256_ROWS * TN256_ROWS;
        int rc = svla_gemm_tn_f32acc(dY, ldy, X, ldx, dW, ldw, db, mb, N, K, stream);
        if (rc) return rc;
        g_force_small_tile = 1;
        rc = svla_gemm_tn_f32acc(dY + (size_t)mb * ldy, ldy, X + (size_t)mb * ldx, ldx, dW, ldw, db, M - mb, N, K, stream);
        g_force_small_tile = 2;
        return rc;
    }
    if ((N % 256) == 0 && (K % 256) == 0 && (M % TN256_ROWS) == 0 && (M >= 16384 || (M >= 8192 && (long)N * K >= 1024L * 512) || g_force_small_tile == 2) && g_force_small_tile != 1) {
        // measured (r03): at 16 k rows the 256-tile kernel is 1.0x (512 x 512) to 2.0x (1536 x 512) the 128-tile one, at 8 k rows 0.7x / 1.2-1.9x
        const int ntile256 = (N / 256) * (K / 256);
        int chunks = 256 / ntile256;                                  // <= one workgroup per CU (no second dispatch wave)
        if (chunks < 1) chunks = 1;
        int chunk_rows = ((M + chunks - 1) / chunks + TN256_ROWS - 1) / TN256_ROWS * TN256_ROWS;
        chunks = (M + chunk_rows - 1) / chunk_rows;
        GemmTn256Args q{dY, ldy, X, ldx, dW, ldw, db, M, N, K, chunk_rows, g_dbg, g_svla_det};
        const size_t lds256 = (size_t)TN_NS * 2 * TN256_ROWS * 256 * sizeof(bf16_t);   // 128 KiB
        static bool attr256 = false;
        if (!attr256) {
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_tn256_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
            attr256 = true;
        }
        if (!(g_dbg & (128 | 8192)) && !g_svla_det.i64[0] && !g_svla_det.i64[1] && svla_asm_has("svla_tn_os")) {
            // output-stationary assembly kernel (asmgen/tn_os_gen.py): 4 waves x 128 x 128 accumulators, 4-slot LDS-DMA ring
            struct { const void* dY; long ldy; const void* X; long ldx; float* dW; long ldw; float* db; int M, N, K, chunk_rows, ntile, ntk, grid, pad; } k =
                {dY, ldy, X, ldx, dW, ldw, db, M, N, K, chunk_rows, ntile256, K / 256, ntile256 * chunks, 0};
            static_assert(sizeof(k) == 88, "kernarg layout of svla_tn_os");
            gemm_log("svla_tn_os", M, N, K);
            return svla_asm_launch("svla_tn_os", &k, sizeof(k), ntile256 * chunks, 256, (hipStream_t)stream);
        }
        if (!(g_dbg & 128)) {
            static bool attr8p = false;
            if (!attr8p) {
                HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_tn8p_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
                attr8p = true;
            }
            gemm_log("gemm_tn8p_bf16_kernel", M, N, K);
            hipLaunchKernelGGL(gemm_tn8p_bf16_kernel, dim3(ntile256 * chunks), dim3(NT256_THREADS), lds256, (hipStream_t)stream, q);
            return svla_launch_status();
        }
        hipLaunchKernelGGL(gemm_tn256_bf16_kernel, dim3(ntile256 * chunks), dim3(NT256_THREADS), lds256, (hipStream_t)stream, q);
        return svla_launch_status();
    }
    if (db) {   // small-tile path: bias gradient as a separate column-sum pass
        const int rc = svla_colsum_bf16(dY, ldy, M, N, 1, db, stream);
        if (rc) return rc;
    }
    const int ntile = (N / 128) * (K / 128);
    // aim for ~2048 workgroups; chunk is a multiple of the 64-row reduction tile
    int chunks = (2048 + ntile - 1) / ntile;
    int chunk_rows = ((M + chunks - 1) / chunks + TK - 1) / TK * TK;
    if (chunk_rows < 4 * TK) chunk_rows = 4 * TK;
    chunks = (M + chunk_rows - 1) / chunk_rows;
    GemmTnArgs p{dY, ldy, X, ldx, dW, ldw, M, N, K, chunk_rows, g_svla_det};
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
                                   float* __restrict__ db, DetCfg det) {
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
        grad_add(det, &db[col], t);
    }
}

// row m is read at memory row m*row_stride (row_stride > 1: e.g. token 0 of every [S, D] group)
extern "C" int svla_colsum_bf16(const bf16_t* dY, long ldy, int M, int N, int row_stride, float* db, void* stream) {
    if (M <= 0 || N <= 0 || (N % 8) || (ldy % 8)) return SVLA_EINVAL;
    const int threads = 256;
    for (int c0 = 0; c0 < N; c0 += 2048) {        // one launch per 2048 columns (256 threads x 8): only the fused QKV gradient of the 768-wide presets (N = 2304) needs two
        const int n = N - c0 < 2048 ? N - c0 : 2048;
        const int rpp = threads / (n / 8);
        int blocks = (M + rpp - 1) / rpp;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(colsum_bf16_kernel, dim3(blocks), dim3(threads), threads * 8 * sizeof(float), (hipStream_t)stream, dY + c0, ldy,
                           M, n, row_stride > 0 ? row_stride : 1, db + c0, g_svla_det);
    }
    return svla_launch_status();
}
