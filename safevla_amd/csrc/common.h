// Common device helpers for the gfx950 (CDNA4 / MI355X) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "launch.h"

#define SVLA_OK 0
#define SVLA_EINVAL (-1)

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (8 bf16 = 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even: native v_cvt_pk_bf16_f32 on gfx950 (the compiler emits it for __bf16 casts)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_hw));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// LDS transpose read (gfx950 ds_read_b64_tr_b16): within each 16-lane group, lane p supplies the address of
// 4 contiguous bf16 = (row p>>2, column quad p&3) of a [4][16] block; lane i receives column i of that block
// (elements row 0..3).  See MI355X guide section 2 / T10.
__device__ __forceinline__ bf16x4 lds_tr16_b64(const bf16_t* p) {
    typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 v4bf;
    v4bf r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) v4bf*)p);
    return __builtin_bit_cast(bf16x4, r);
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_hw;
// D(32x32) += A(32x16) . B(16x32): lane l holds A[l&31][8*(l>>5)+0..7], B[8*(l>>5)+0..7][l&31];
// D[reg]: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
// D(16x16) += A(16x32) . B(32x16): lane l holds A[l&15][8*(l>>4)+0..7], B[8*(l>>4)+0..7][l&15];
// D[reg]: col = l&15, row = 4*(l>>4) + reg
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}

#define HIP_CHECK_RET(expr)                        \
    do {                                           \
        hipError_t _e = (expr);                    \
        if (_e != hipSuccess) return (int)_e;      \
    } while (0)

static inline int svla_launch_status() { return (int)hipGetLastError(); }

// ---- counter-based dropout (definition in include/svla.h: svla_dropout) ----------------------------------------------------
struct svla_dropout { unsigned seed, stream; float p; int row_mult; const unsigned* seed_dev; };
struct DropCfg { unsigned key, thr; float scale; int row_mult; const unsigned* seed_dev; unsigned stream_key; };   // thr == 0: off
__host__ __device__ inline DropCfg drop_cfg(const svla_dropout* d) {
    DropCfg c{0u, 0u, 1.f, 1, nullptr, 0u};
    if (d && d->p > 0.f) {
        c.stream_key = d->stream * 0xC2B2AE3Du;
        c.key = d->seed ^ c.stream_key;
        c.seed_dev = d->seed_dev;
        c.thr = (unsigned)(d->p * 65536.f + 0.5f);
        c.scale = 1.f / (1.f - d->p);
        c.row_mult = d->row_mult > 0 ? d->row_mult : 1;
    }
    return c;
}
// device-resident pass seed (HIP-graph replays): resolve once at kernel start
__device__ __forceinline__ DropCfg drop_resolve(DropCfg c) {
    if (c.thr && c.seed_dev) c.key = *c.seed_dev ^ c.stream_key;
    return c;
}
// mixer of the dropout hash: two xorshift-multiply rounds on 24-bit multiplies (v_mul_u32_u24 is full rate on CDNA, the 32-bit
// v_mul_lo_u32 of the usual lowbias32 quarter rate -- the hash is the VALU hot spot of every train-mode epilogue and of the attention
// kernels).  umul24(x, K) = low 32 bits of (x & 0xffffff) * K, K < 2^24.
__device__ __forceinline__ unsigned drop_mix(unsigned x) {
    x ^= x >> 16; x = __umul24(x, 0xEB352Du); x ^= x >> 13; x = __umul24(x, 0x6CA68Bu); x ^= x >> 16;
    return x;
}
// 32 random bits for the element pair (e >> 1): low half -> even element, high half -> odd element
__device__ __forceinline__ unsigned drop_bits(unsigned key, unsigned long long pair) {
    unsigned x = ((unsigned)pair * 0x9E3779B1u) ^ ((unsigned)(pair >> 32) * 0x85EBCA77u) ^ key;
    return drop_mix(x);
}
// keep-mask (bit i <=> element e0 + i kept) of 4 consecutive elements, e0 % 4 == 0
__device__ __forceinline__ unsigned drop_keep4(const DropCfg& c, unsigned long long e0) {
    const unsigned r0 = drop_bits(c.key, e0 >> 1), r1 = drop_bits(c.key, (e0 >> 1) + 1);
    return ((r0 & 0xffffu) >= c.thr ? 1u : 0u) | ((r0 >> 16) >= c.thr ? 2u : 0u) | ((r1 & 0xffffu) >= c.thr ? 4u : 0u) | ((r1 >> 16) >= c.thr ? 8u : 0u);
}


// ---- deterministic gradient accumulation (svla_det_config) -------------------------------------------------------------------
// fp32 atomicAdd makes every accumulated gradient depend on the arrival order of the workgroups.  With a registered shadow buffer the
// same partial sums are added as 64-bit FIXED-POINT integers (2^-52 resolution: the gradients are 1/n_total-scaled, per-workgroup partials
// of 1e-10 .. 1e-12 are common, and a 2^-40 grid flushed the smallest of them -- ADVICE r3): integer addition is associative, so the
// result is bitwise repeatable; each partial is rounded once to the 2^-52 grid (relative error < 2^-24 for every partial above 1e-9,
// absolute 1.1e-16 below), no rounding between the adds.  RANGE (numbers for the default grid b = 52): the int64 holds +-2048; a partial enters the shadow only if
// |partial| < max_partial = 0.25, so a sum of up to 8192 partials (the largest grid that accumulates into one element is norm_bwd's
// 4096 workgroups) cannot wrap.  Larger and non-finite partials -- a diverged backward -- bypass the shadow and go to the fp32 buffer with a
// plain atomic (not repeatable, but visible as NaN / Inf / a huge gradient instead of a silently wrapped one); the per-block LDS table of
// decoder_embed_bwd applies the same rule.  svla_det_finalize folds the shadow back into the fp32 buffer.  Up to two registered ranges: the
// flat gradient buffer, and a scratch range for accumulated intermediates.
struct DetCfg { float* f32[2]; long long* i64[2]; long n[2]; unsigned long long* bypass; float scale, max_partial; double unscale; };
// bypass: device counter of partials that had a shadow but left it (svla_det_bypass_count).  scale = 2^b, unscale = 2^-b, max_partial = 2^(50 - b): the grid of the shadow
// (svla_det_set_grid; b = 52 by default).  8192 partials of magnitude < 2^(50 - b) sum to < 2^63 grid units: the int64 cannot wrap.  The engine lowers b for small
// minibatches, whose 1 / n_total-scaled partials are larger (64 rows: b = 44, partials up to 64) -- round 6: with the fixed b = 52 a 64-row update sent 5 964 partials >= 0.25
// around the shadow (tests/test_engine_fullsize_gpu.py found it through the new counter).
extern DetCfg g_svla_det;                 // host-side current configuration (misc.hip); all-null = plain fp32 atomics
__device__ __forceinline__ unsigned long long det_fixed(const DetCfg& d, float v) { return (unsigned long long)__float2ll_rn(v * d.scale); }
__device__ __forceinline__ void grad_add(const DetCfg& d, float* p, float v) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (d.i64[k]) {
            const long off = p - d.f32[k];
            if (off >= 0 && off < d.n[k]) {
                if (fabsf(v) < d.max_partial) { atomicAdd((unsigned long long*)(d.i64[k] + off), det_fixed(d, v)); return; }
                // NaN / Inf / out-of-range: the fp32 atomic below (visible, not repeatable) -- and counted, so that a "deterministic" run that was not says so
                if (d.bypass) atomicAdd(d.bypass, 1ull);
                break;
            }
        }
    }
    atomicAdd(p, v);
}
