// Fused softmax attention, forward + backward, bf16 in/out, fp32 softmax/accumulators, head_dim 64
// (gfx950, v_mfma_f32_16x16x32_bf16).  Sequences on this path are short (fusion transformer S = 1+84+84+L <= 256,
// rollout decoder S = T <= 256, T5 L <= 64, ViT 433), so one workgroup owns one (batch row, head) with K and V
// resident in LDS, and the full score row of a 16-query tile lives in registers: exact two-pass softmax, no online
// rescale.  Reference ops: nn.MultiheadAttention inside nn.TransformerEncoderLayer (no mask;
// allenact_dino_transformer.py:545-552,702-708), llama Attention with the block-causal ``traj_index`` mask
// (llama/model.py:249-322 + allenact_dino_transformer.py:398-402), T5 self-attention (additive position bias +
// key padding mask, no 1/sqrt(d) scale).
//
// Layout trick ("swapped QK^T"): S^T = K.Q^T puts the query index on lane&15, so a query's whole score row is
// lane-local (+2 cross-lane steps), and the probabilities already sit in the MFMA A-operand layout of P.V;
// V (and K/Q/dO in the backward) are read as B operands straight from row-major LDS with ds_read_b64_tr_b16.
#include "common.h"

#ifndef ATT_SWZ_OLD
#define ATT_SWZ_OLD 0
#endif
#ifndef ATT_EARLY_TR
#define ATT_EARLY_TR 0      // measured: no gain at 3 workgroups per CU (other waves hide the LDS latency), spills with dropout
#endif
#define HD 64
#define LDSROW 64   // LDS row = 128 B (one head slice row), 16-byte chunks XOR-swizzled: 48 KiB for two [192, 64] operands,
                    // so THREE workgroups fit a CU's 160 KiB (the kernels are latency-bound: 1 -> 2 workgroups/CU = 1.74x)
#define ATT_THREADS 256

enum { MASK_NONE = 0, MASK_BLOCK_CAUSAL = 1 };

struct AttnArgs {
    const bf16_t *Q, *K, *V; long ld;     // token row stride (elements) of the q/k/v tensors
    bf16_t* O; long ldo;
    float* LSE;                           // [rows, H, S]
    const bf16_t* dO; long lddo;
    bf16_t *dQ, *dK, *dV; long ldd;
    const int* traj;                      // [rows, S] (MASK_BLOCK_CAUSAL)
    const float* bias;                    // [H, S, S] additive (T5) or null
    const unsigned char* kvalid;          // [rows, S] key padding mask or null
    int S, H, mask_mode;
    float scale;
    int kv_rows;                          // K/V (dK/dV) token rows allocated per batch row (>= S; KV caches), default S
    int Sq;                               // query rows per batch row present in Q / O / dO / dQ / LSE (<= S; S = all)
    long ldq, lddq;                       // token row strides of Q and dQ (the K/V tensors use ld / ldd)
    float* Dws;                           // [rows, H, Sq] rowsum(dO * O): written by the dQ kernel, read by the dK/dV kernel (or null)
    DropCfg drop;                         // dropout on the attention probabilities (thr == 0: off)
    int xcd_rows;                         // single-pass backward: workgroup -> (row, head) so that every XCD walks whole rows (0: row-major items)
};

// element index of probability (row r, head h, query q, key k): ((r*H + h)*S + q) * SP4 + k with the key stride SP4 = S rounded
// up to a multiple of 4, so that the 4 consecutive keys a lane holds share two RNG words (include/svla.h: svla_dropout)
__device__ __forceinline__ unsigned long long att_drop_row(const AttnArgs& p, int r, int h, int q) {
    return ((unsigned long long)((size_t)r * p.H + h) * p.S + q) * (unsigned long long)((p.S + 3) & ~3);
}
__device__ __forceinline__ bool att_keep1(const DropCfg& c, unsigned long long e) {
    const unsigned x = drop_bits(c.key, e >> 1);
    return ((e & 1) ? (x >> 16) : (x & 0xffffu)) >= c.thr;
}

// physical 16-byte chunk of logical chunk c in row r: c ^ f(x), x = (r >> 1) & 7 (64 banks x 4 B; a row is 32 banks, so the row
// parity picks the bank half and f only has to spread the eight values of x over the eight 16-byte slots of a half).  Three
// access patterns constrain f:
//  * ds_read_b128 row fragments (row = lane & 15, logical chunk = lane >> 4): the hardware serves lanes {0-3,12-15,20-27},
//    {4-11,16-19,28-31}, ... as groups, i.e. rows with x in {0,1,6,7} on chunk c together with rows with x in {2,3,4,5} on
//    chunk c ^ 1: conflict-free iff f is a permutation and f({2,3,4,5}) is a union of two chunk pairs {2k, 2k+1};
//  * ds_read_b64_tr_b16 (32 lanes per pass = 8 rows x 32 B = one chunk PAIR per row): f >> 1 must be distinct over
//    x = 0..3 and over x = 4..7.  (f = x, the first layout, put rows 2,3 on the chunk pair of rows 0,1: every transposed
//    read was a 2-way conflict.)
//  * staging stores (8 lanes = one row): any f.
// f = 0,2,4,6,5,7,1,3 satisfies all three.
__device__ __forceinline__ int att_swz(int row) {
    const int x = (row >> 1) & 7;
    return ATT_SWZ_OLD ? x : ((((x + ((x >> 2) << 1)) & 3) << 1) | (x >> 2));
}

// Lane bases of the two access patterns (tile rows are multiples of 16, so the swizzle term depends on the lane only and the
// tile offset stays a compile-time immediate of the ds_read):
//   row fragment  : row = tile + (lane & 15), logical chunk (lane >> 4) [+4 for columns 32..63]
//   transposed    : row = tile + 4 (lane >> 4) + ((lane & 15) >> 2), columns dt*16 + 4 ((lane & 15) & 3) .. +3
struct RowBase { const bf16_t* lo; const bf16_t* hi; };
__device__ __forceinline__ RowBase att_row_base(const bf16_t* tile, int lane) {
    const int ql = lane & 15, g = lane >> 4, f = att_swz(ql);
    return RowBase{tile + ql * LDSROW + ((g ^ f) << 3), tile + ql * LDSROW + (((g + 4) ^ f) << 3)};
}
struct TrBase { const bf16_t* d[4]; };
__device__ __forceinline__ TrBase att_tr_base(const bf16_t* tile, int lane) {
    const int ql = lane & 15, g = lane >> 4;
    const int row = 4 * g + (ql >> 2), f = att_swz(row);
    TrBase t;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) t.d[dt] = tile + row * LDSROW + (((2 * dt + ((ql & 3) >> 1)) ^ f) << 3) + 4 * (ql & 1);
    return t;
}
__device__ __forceinline__ bf16x8 lds_row8i(const bf16_t* lane_base, int tile_row0) {
    return *(const bf16x8*)(lane_base + tile_row0 * LDSROW);
}
// B/A-operand gather: 8 reduction slots = rows {rA + 4g + 0..3, rB + 4g + 0..3}, column dt*16 + (lane & 15)
__device__ __forceinline__ bf16x8 lds_tr8i(const bf16_t* lane_base_dt, int rA, int rB) {
    const bf16x4 lo = lds_tr16_b64(lane_base_dt + rA * LDSROW);
    const bf16x4 hi = lds_tr16_b64(lane_base_dt + rB * LDSROW);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ bf16x8 pack8(const float (&v)[8]) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
    return __builtin_bit_cast(bf16x8, w);
}
#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

// stage an [S, 64] head slice into LDS rows (zero-filled up to SP rows): all global loads are issued before the first LDS
// store so the HBM/L2 latency is paid once, not once per loop trip
template <int SP, int THREADS = ATT_THREADS>
__device__ __forceinline__ void stage_head(bf16_t* dst, const bf16_t* src, long ld, int S, int tid) {
    constexpr int IT = SP * 8 / THREADS;       // SP is a multiple of 32 (of 64 with 512 threads)
    u32x4 w[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int q = tid + i * THREADS;
        const int row = q >> 3, c = q & 7;
        w[i] = u32x4{0, 0, 0, 0};
        if (row < S) w[i] = *(const u32x4*)(src + (size_t)row * ld + c * 8);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int q = tid + i * THREADS;
        *(u32x4*)(dst + (q >> 3) * LDSROW + (((q & 7) ^ att_swz(q >> 3)) << 3)) = w[i];
    }
}

__device__ __forceinline__ bool masked(const AttnArgs& p, int q, int key, const int* traj_s, const unsigned char* kv_s) {
    if (key >= p.S) return true;
    if (p.mask_mode == MASK_BLOCK_CAUSAL && (key > q || traj_s[key] != traj_s[q])) return true;
    if (kv_s && !kv_s[key]) return true;
    return false;
}

// ================================================================================================ forward
// GENERIC = false: no bias / traj / padding mask (the fusion encoder, >99 % of the attention work): only the ragged tail
// of the last key tile is masked and the softmax runs in the exp2 domain with the scale folded in.
// NW = waves per workgroup.  The long-sequence shapes (S > 256: the ViT's 433 tokens) need > 80 KiB of LDS for K and V, i.e. ONE workgroup
// per CU: eight waves instead of four share that K/V image (2 waves per SIMD hide each other's latency; measured 1.5x on S = 433).
// EXACT (round 5; the ViT's S = 433 with NKT = 28, S = 257 with NKT = 18): S > (NKT - 2) * 16, so only the LAST TWO key tiles can hold padded keys (zero rows
// in LDS, masked to -inf) -- the per-tile runtime predicates of the general form (28 unrolled wave-uniform conditions) cost 316 spilled SGPRs through
// v_readlane / v_writelane.
template <int NKT, bool GENERIC, int NW = 4, bool EXACT = false>
__device__ __forceinline__ void attn_fwd_kernel_body(AttnArgs p) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 16;
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + SP * LDSROW;
    int* traj_s = (int*)(Vs + SP * LDSROW);
    unsigned char* kv_s = (unsigned char*)(traj_s + SP);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const size_t tok0 = (size_t)r * p.kv_rows, mtok0 = (size_t)r * p.S;
    const int S = p.S;
    stage_head<SP, NW * 64>(Ks, p.K + tok0 * p.ld + h * HD, p.ld, S, tid);
    stage_head<SP, NW * 64>(Vs, p.V + tok0 * p.ld + h * HD, p.ld, S, tid);
    for (int i = tid; i < SP; i += NW * 64) {
        traj_s[i] = (p.traj && i < S) ? p.traj[mtok0 + i] : -1;
        kv_s[i] = (p.kvalid && i < S) ? p.kvalid[mtok0 + i] : 1;
    }
    const int ql = lane & 15, g = lane >> 4;
    const int Sq = p.Sq;
    const size_t qtok0 = (size_t)r * Sq;
    const int nqt = (Sq + 15) / 16;
    // Q fragments of all of this wave's query tiles (qt = wid, wid+NW, ...): issued before the barrier so their latency
    // overlaps the K/V staging instead of stalling every tile
    constexpr int MAXQT = (NKT + NW - 1) / NW;
    bf16x8 qall[MAXQT][2];
#pragma unroll
    for (int t = 0; t < MAXQT; ++t) {
        const int q = (wid + NW * t) * 16 + ql;
        const bool ok = q < Sq;
        const bf16_t* qp = p.Q + (qtok0 + (ok ? q : 0)) * p.ldq + h * HD + 8 * g;
        qall[t][0] = ok ? *(const bf16x8*)qp : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        qall[t][1] = ok ? *(const bf16x8*)(qp + 32) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    __syncthreads();
    const unsigned char* kvp = p.kvalid ? kv_s : nullptr;
    const RowBase Krow = att_row_base(Ks, lane);
    const TrBase Vtr = att_tr_base(Vs, lane);
#pragma unroll
    for (int t = 0; t < MAXQT; ++t) {
        const int qt = wid + NW * t;
        if (qt >= nqt) break;
        const int q = qt * 16 + ql;
        const bf16x8 qf[2] = {qall[t][0], qall[t][1]};
        float sc[NKT][4];
        float mx = -INFINITY;
        const float sl2 = p.scale * LOG2E;      // scores are kept in the log2 domain: p = exp2(s*scale*log2e - max)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if (EXACT || kt * 16 < S) {
                a = mfma16(lds_row8i(Krow.lo, kt * 16), qf[0], a);
                a = mfma16(lds_row8i(Krow.hi, kt * 16), qf[1], a);
            }
            if constexpr (GENERIC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kt * 16 + 4 * g + e;
                    float s = a[e] * p.scale;
                    if (p.bias && q < Sq && key < S) s += p.bias[((size_t)h * S + q) * S + key];
                    s *= LOG2E;
                    if (masked(p, q < Sq ? q : 0, key, traj_s, kvp)) s = -INFINITY;
                    sc[kt][e] = s;
                    mx = fmaxf(mx, s);
                }
            } else {
                const bool tail = EXACT ? kt >= NKT - 2 : (kt + 1) * 16 > S;      // wave-uniform: only the last (ragged) key tiles need masking
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = a[e] * sl2;
                    if (tail && kt * 16 + 4 * g + e >= S) s = -INFINITY;
                    sc[kt][e] = s;
                    mx = fmaxf(mx, s);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (mx == -INFINITY) mx = 0.f;
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[kt][e] = __builtin_amdgcn_exp2f(sc[kt][e] - mx); lsum += sc[kt][e]; }
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        if (p.drop.thr) {      // dropout on the normalised probabilities: zero here, 1/(1-p) folded into the final scale
            const unsigned long long rb = att_drop_row(p, r, h, q < Sq ? q : 0);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const unsigned keep = drop_keep4(p.drop, rb + kt * 16 + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!((keep >> e) & 1u)) sc[kt][e] = 0.f;
            }
        }
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NKT / 2; ++u) {
            if (EXACT || u * 32 < S) {
                const float pv[8] = {sc[2 * u][0], sc[2 * u][1], sc[2 * u][2], sc[2 * u][3],
                                     sc[2 * u + 1][0], sc[2 * u + 1][1], sc[2 * u + 1][2], sc[2 * u + 1][3]};
                const bf16x8 pa = pack8(pv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = mfma16(lds_tr8i(Vtr.d[dt], 32 * u, 32 * u + 16), pa, o[dt]);   // O^T: rows = head dims, cols = queries
            }
        }
        // o[dt][e]: query q (this lane's own softmax row), head dim dt*16 + 4g + e  ->  8-byte stores, 32 B per row per dt
        const float inv = lsum > 0.f ? p.drop.scale / lsum : 0.f;
        if (q < Sq) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x2 w = {pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv)};
                *(u32x2*)(p.O + (qtok0 + q) * p.ldo + h * HD + dt * 16 + 4 * g) = w;
            }
        }
        if (p.LSE && g == 0 && q < Sq) p.LSE[((size_t)r * p.H + h) * Sq + q] = (mx + __log2f(lsum)) * LN2;   // natural-log LSE
    }
}
// (no same-named __global__: every launch goes through the grouped form, a single launch being a group of one -- SVLA_LAUNCH_TWIN, csrc/launch.h)

// item -> (row, head) with every XCD walking whole rows: item i is processed on XCD i % 8 (workgroup b on XCD b % 8; the persistent forward's grid is a multiple of 8),
// so items 8 j + x, j = 8 g .. 8 g + 7, become the eight heads of row 8 g + x: the eight 128-byte head slices of a token's line group go through ONE L2
__device__ __forceinline__ void att_xcd_item(const AttnArgs& p, int item, int nitems, int& r, int& h) {
    if (!p.xcd_rows) return;
    const int x = item & 7, j = item >> 3, nfull = (nitems / 64) * 8;
    if ((j >> 3) * 8 + x < nfull) { r = (j >> 3) * 8 + x; h = j & 7; }
}
// per-(row, head) constants of the incremental dropout hash (used by the persistent forward and the single-pass backward)
struct AttDrop {               // per (row, head): hash of pair index P0 + delta, delta < 2^16
    unsigned a0;               // lo(P0) * C1
    unsigned hb;               // (hi(P0) * C2) ^ key
    bool wrap;                 // lo(P0) + delta may carry into hi(P): take the generic path (never for < 2^33 probabilities per tensor)
    unsigned long long p0;
};
__device__ __forceinline__ AttDrop att_drop_head(const AttnArgs& p, int r, int h) {
    AttDrop d;
    d.p0 = att_drop_row(p, r, h, 0) >> 1;
    d.a0 = (unsigned)d.p0 * 0x9E3779B1u;
    d.hb = ((unsigned)(d.p0 >> 32) * 0x85EBCA77u) ^ p.drop.key;
    d.wrap = (unsigned)d.p0 > 0xFFFF0000u;
    return d;
}

// ------------------------------------------------------------------------------------------------ persistent forward
// The one-item-per-workgroup kernel above is latency-bound (stage -> barrier -> compute -> store; 1 -> 2 workgroups per CU
// = 1.74x).  For the shape that carries > 99 % of the attention work (no mask / bias, S <= 192) a workgroup walks a list of
// (row, head) items instead and keeps the NEXT item's K/V head slices and Q fragments in flight in registers (72 VGPRs)
// while it computes the current item from LDS: global latency is off the critical path after the first item.
template <int NKT>
__device__ __forceinline__ void attn_fwd_persist_kernel_body(AttnArgs p, int nitems) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 16, IT = SP * 8 / ATT_THREADS, MAXQT = (NKT + 3) / 4;
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + SP * LDSROW;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ql = lane & 15, g = lane >> 4;
    const int S = p.S, Sq = p.Sq, nqt = (Sq + 15) / 16;
    u32x4 kreg[IT], vreg[IT];
    bf16x8 qreg[MAXQT][2];
    auto fetch = [&](int item) {
        int r = item / p.H, h = item % p.H;
        att_xcd_item(p, item, nitems, r, h);
        const bf16_t* kp = p.K + (size_t)r * p.kv_rows * p.ld + h * HD;
        const bf16_t* vp = p.V + (size_t)r * p.kv_rows * p.ld + h * HD;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int q = tid + i * ATT_THREADS, row = q >> 3, c = q & 7;
            kreg[i] = u32x4{0, 0, 0, 0}; vreg[i] = u32x4{0, 0, 0, 0};
            if (row < S) {
                kreg[i] = *(const u32x4*)(kp + (size_t)row * p.ld + c * 8);
                vreg[i] = *(const u32x4*)(vp + (size_t)row * p.ld + c * 8);
            }
        }
#pragma unroll
        for (int t = 0; t < MAXQT; ++t) {
            const int q = (wid + 4 * t) * 16 + ql;
            const bool ok = q < Sq;
            const bf16_t* qp = p.Q + ((size_t)r * Sq + (ok ? q : 0)) * p.ldq + h * HD + 8 * g;
            qreg[t][0] = ok ? *(const bf16x8*)qp : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            qreg[t][1] = ok ? *(const bf16x8*)(qp + 32) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    };
    int item = blockIdx.x;
    if (item >= nitems) return;
    fetch(item);
    const RowBase Krow = att_row_base(Ks, lane);
    const TrBase Vtr = att_tr_base(Vs, lane);
    const float sl2 = p.scale * LOG2E;      // scores are kept in the log2 domain: p = exp2(s*scale*log2e - max)
    for (; item < nitems; item += gridDim.x) {
        int r = item / p.H, h = item % p.H;
        att_xcd_item(p, item, nitems, r, h);
        const size_t qtok0 = (size_t)r * Sq;
        __syncthreads();                     // every wave is done reading the previous item's K/V
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int q = tid + i * ATT_THREADS;
            const int off = (q >> 3) * LDSROW + (((q & 7) ^ att_swz(q >> 3)) << 3);
            *(u32x4*)(Ks + off) = kreg[i];
            *(u32x4*)(Vs + off) = vreg[i];
        }
        bf16x8 qcur[MAXQT][2];
#pragma unroll
        for (int t = 0; t < MAXQT; ++t) { qcur[t][0] = qreg[t][0]; qcur[t][1] = qreg[t][1]; }
        __syncthreads();
        if (item + (int)gridDim.x < nitems) fetch(item + gridDim.x);
#pragma unroll
        for (int t = 0; t < MAXQT; ++t) {
            const int qt = wid + 4 * t;
            if (qt >= nqt) break;
            const int q = qt * 16 + ql;
            // raw scores stay in the MFMA result vectors; scale (> 0) and log2(e) are folded into the exp2 argument:
            //   p = exp2(raw * sl2 - max(raw) * sl2)        (vector arithmetic so the compiler can use packed fp32 ops)
            f32x4 sc[NKT];
            f32x4 mx4 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                a = mfma16(lds_row8i(Krow.lo, kt * 16), qcur[t][0], a);
                a = mfma16(lds_row8i(Krow.hi, kt * 16), qcur[t][1], a);
                if (kt >= NKT - 2) {   // (NKT-2)*16 < S <= NKT*16 (launcher): only the last two key tiles can hold padded keys
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kt * 16 + 4 * g + e >= S) a[e] = -INFINITY;
                }
                sc[kt] = a;
                mx4 = __builtin_elementwise_max(mx4, a);
            }
            float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float nm = -mx * sl2;
            const f32x4 nm4 = {nm, nm, nm, nm}, sl4 = {sl2, sl2, sl2, sl2};
            f32x4 ls4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const f32x4 x = sc[kt] * sl4 + nm4;
#pragma unroll
                for (int e = 0; e < 4; ++e) sc[kt][e] = __builtin_amdgcn_exp2f(x[e]);
                ls4 += sc[kt];
            }
            float lsum = (ls4[0] + ls4[1]) + (ls4[2] + ls4[3]);
            lsum += __shfl_xor(lsum, 16, 64);
            lsum += __shfl_xor(lsum, 32, 64);
            if (p.drop.thr) {      // wave-uniform; dropout on the normalised probabilities (scale folded into ``inv``)
                // a lane holds 4 consecutive keys (kt*16 + 4g + e) of one query = two RNG words; the pair index P0 + q*(S4/2) + key/2 is linear in (q, key):
                // lo(P)*C1 = lane constant + wave-uniform term (the single-pass backward regenerates the same words the same way)
                const AttDrop dh = att_drop_head(p, r, h);
                const int S4 = (S + 3) & ~3, hS = S4 >> 1;
                const unsigned thr = p.drop.thr;
                if (!dh.wrap) {
                    const unsigned al = dh.a0 + (unsigned)(ql * hS + 2 * g) * 0x9E3779B1u;
                    const unsigned uq = (unsigned)(qt * 16 * hS) * 0x9E3779B1u;
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) {
                        const unsigned ub = uq + (unsigned)(kt * 8) * 0x9E3779B1u;
                        const unsigned r0 = drop_mix((al + ub) ^ dh.hb), r1 = drop_mix((al + ub + 0x9E3779B1u) ^ dh.hb);
                        if ((r0 & 0xffffu) < thr) sc[kt][0] = 0.f;
                        if ((r0 >> 16) < thr) sc[kt][1] = 0.f;
                        if ((r1 & 0xffffu) < thr) sc[kt][2] = 0.f;
                        if ((r1 >> 16) < thr) sc[kt][3] = 0.f;
                    }
                } else {
                    const unsigned long long rb = att_drop_row(p, r, h, q < Sq ? q : 0);
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) {
                        const unsigned keep = drop_keep4(p.drop, rb + kt * 16 + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (!((keep >> e) & 1u)) sc[kt][e] = 0.f;
                    }
                }
            }
            f32x4 o[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NKT / 2; ++u) {
                const float pv[8] = {sc[2 * u][0], sc[2 * u][1], sc[2 * u][2], sc[2 * u][3],
                                     sc[2 * u + 1][0], sc[2 * u + 1][1], sc[2 * u + 1][2], sc[2 * u + 1][3]};
                const bf16x8 pa = pack8(pv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    o[dt] = mfma16(lds_tr8i(Vtr.d[dt], 32 * u, 32 * u + 16), pa, o[dt]);
            }
            const float inv = lsum > 0.f ? p.drop.scale / lsum : 0.f;
            if (q < Sq) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const u32x2 w = {pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv)};
                    *(u32x2*)(p.O + (qtok0 + q) * p.ldo + h * HD + dt * 16 + 4 * g) = w;
                }
            }
            if (p.LSE && g == 0 && q < Sq) p.LSE[((size_t)r * p.H + h) * Sq + q] = (mx * sl2 + __log2f(lsum)) * LN2;
        }
    }
}
// (launched as svla_grouped<&attn_fwd_persist_kernel_body<NKT>, ATT_THREADS, 2>: SVLA_LAUNCH_TWIN)

// ================================================================================================ backward
// Two kernels, each with only two [S,64] operands resident in LDS (2 workgroups per CU):
//   dQ  kernel (waves own query tiles, swapped layout, K and V in LDS):   dQ = dS.K
//   dKV kernel (waves own key tiles, Q and dO in LDS):                    dK = dS^T.Q, dV = P^T.dO
// P is recomputed from the saved log-sum-exp; D = rowsum(dO * O).
__device__ __forceinline__ bf16x8 gld8(const bf16_t* p, bool ok) {
    return ok ? *(const bf16x8*)p : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
}
__device__ __forceinline__ float dot8(bf16x8 a, bf16x8 b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += bf2f((bf16_t)a[i]) * bf2f((bf16_t)b[i]);
    return s;
}

template <int NKT, bool GENERIC>
__global__ void __launch_bounds__(ATT_THREADS) attn_bwd_dq_kernel(AttnArgs p) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 16;
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + SP * LDSROW;
    int* traj_s = (int*)(Vs + SP * LDSROW);
    unsigned char* kv_s = (unsigned char*)(traj_s + SP);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const size_t tok0 = (size_t)r * p.S;
    const int S = p.S;
    stage_head<SP>(Ks, p.K + tok0 * p.ld + h * HD, p.ld, S, tid);
    stage_head<SP>(Vs, p.V + tok0 * p.ld + h * HD, p.ld, S, tid);
    for (int i = tid; i < SP; i += ATT_THREADS) {
        traj_s[i] = (p.traj && i < S) ? p.traj[tok0 + i] : -1;
        kv_s[i] = (p.kvalid && i < S) ? p.kvalid[tok0 + i] : 1;
    }
    const int ql = lane & 15, g = lane >> 4;
    const int Sq = p.Sq;
    const size_t qtok0 = (size_t)r * Sq;
    // Q / dO fragments, D partials and LSE of all of this wave's query tiles, issued ahead of the staging barrier
    constexpr int MAXQT = (NKT + 3) / 4;
    bf16x8 qall[MAXQT][4];
    float dall[MAXQT], lall[MAXQT];
#pragma unroll
    for (int t = 0; t < MAXQT; ++t) {
        const int q = (wid + 4 * t) * 16 + ql;
        const bool qok = q < Sq;
        const bf16_t* qp = p.Q + (qtok0 + (qok ? q : 0)) * p.ldq + h * HD + 8 * g;
        const bf16_t* gp = p.dO + (qtok0 + (qok ? q : 0)) * p.lddo + h * HD + 8 * g;
        const bf16_t* op = p.O + (qtok0 + (qok ? q : 0)) * p.ldo + h * HD + 8 * g;
        qall[t][0] = gld8(qp, qok); qall[t][1] = gld8(qp + 32, qok);
        qall[t][2] = gld8(gp, qok); qall[t][3] = gld8(gp + 32, qok);
        dall[t] = dot8(qall[t][2], gld8(op, qok)) + dot8(qall[t][3], gld8(op + 32, qok));
        lall[t] = qok ? p.LSE[((size_t)r * p.H + h) * Sq + q] : INFINITY;
    }
    __syncthreads();
    const unsigned char* kvp = p.kvalid ? kv_s : nullptr;
    const RowBase Krow = att_row_base(Ks, lane), Vrow = att_row_base(Vs, lane);
    const TrBase Ktr = att_tr_base(Ks, lane);
    const int ntile = (Sq + 15) / 16;
#pragma unroll
    for (int t = 0; t < MAXQT; ++t) {
        const int qt = wid + 4 * t;
        if (qt >= ntile) break;
        const int q = qt * 16 + ql;
        const bool qok = q < Sq;
        const bf16x8 qf0 = qall[t][0], qf1 = qall[t][1], gf0 = qall[t][2], gf1 = qall[t][3];
        float D_q = dall[t];
        D_q += __shfl_xor(D_q, 16, 64);
        D_q += __shfl_xor(D_q, 32, 64);
        const float lse_q = lall[t];
        const float sl2 = p.scale * LOG2E, lse2_q = lse_q * LOG2E;
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NKT / 2; ++u) {
            if (u * 32 < S) {
                float dsv[8];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int kt = 2 * u + e2;
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    s = mfma16(lds_row8i(Krow.lo, kt * 16), qf0, s);
                    s = mfma16(lds_row8i(Krow.hi, kt * 16), qf1, s);
                    dp = mfma16(lds_row8i(Vrow.lo, kt * 16), gf0, dp);
                    dp = mfma16(lds_row8i(Vrow.hi, kt * 16), gf1, dp);
                    // dP = keep/(1-p) * (dO V^T): the forward's keep-mask, regenerated
                    const unsigned dkeep = p.drop.thr ? drop_keep4(p.drop, att_drop_row(p, r, h, qok ? q : 0) + kt * 16 + 4 * g) : 0xfu;
                    if constexpr (GENERIC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int key = kt * 16 + 4 * g + e;
                            float sv = s[e] * p.scale;
                            if (p.bias && qok && key < S) sv += p.bias[((size_t)h * S + q) * S + key];
                            const bool mk = !qok || masked(p, qok ? q : 0, key, traj_s, kvp);
                            const float pr = mk ? 0.f : __expf(sv - lse_q);
                            dsv[e2 * 4 + e] = pr * (((dkeep >> e) & 1u ? dp[e] * p.drop.scale : 0.f) - D_q) * p.scale;
                        }
                    } else {
                        // no mask needed: padded keys have zero K rows (their dS never reaches dQ), padded queries have
                        // lse = +inf => P = 0
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            dsv[e2 * 4 + e] = __builtin_amdgcn_exp2f(s[e] * sl2 - lse2_q) * (((dkeep >> e) & 1u ? dp[e] * p.drop.scale : 0.f) - D_q) * p.scale;
                    }
                }
                const bf16x8 da = pack8(dsv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    dq[dt] = mfma16(lds_tr8i(Ktr.d[dt], 32 * u, 32 * u + 16), da, dq[dt]);   // dQ^T: rows = head dims, cols = queries
            }
        }
        if (qok) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x2 w = {pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3])};
                *(u32x2*)(p.dQ + (qtok0 + q) * p.lddq + h * HD + dt * 16 + 4 * g) = w;
            }
        }
    }
}

template <int NKT, bool GENERIC>
__global__ void __launch_bounds__(ATT_THREADS, NKT <= 12 ? 3 : 1) attn_bwd_dkv_kernel(AttnArgs p) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 16;
    bf16_t* Qs = (bf16_t*)smem;
    bf16_t* Gs = Qs + SP * LDSROW;  // dO
    float* lse_s = (float*)(Gs + SP * LDSROW);
    float* D_s = lse_s + SP;
    int* traj_s = (int*)(D_s + SP);
    unsigned char* kv_s = (unsigned char*)(traj_s + SP);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const size_t tok0 = (size_t)r * p.S;
    const int S = p.S;
    // this wave's K/V B-operand fragments: the first key tile's are issued ahead of the staging barrier, tile t+1's at the top
    // of tile t (two register sets instead of MAXKT: the kernel must fit 168 VGPRs to run 3 workgroups per CU)
    constexpr int MAXKT = (NKT + 3) / 4;
    bf16x8 kvbuf[2][4];
    auto load_kv = [&](int t, bf16x8 (&dst)[4]) {
        const int keyl = (wid + 4 * t) * 16 + (lane & 15);
        const bool kok = keyl < S;
        const bf16_t* kp = p.K + (tok0 + (kok ? keyl : 0)) * p.ld + h * HD + 8 * (lane >> 4);
        const bf16_t* vp = p.V + (tok0 + (kok ? keyl : 0)) * p.ld + h * HD + 8 * (lane >> 4);
        dst[0] = gld8(kp, kok); dst[1] = gld8(kp + 32, kok);
        dst[2] = gld8(vp, kok); dst[3] = gld8(vp + 32, kok);
    };
    load_kv(0, kvbuf[0]);
    const int Sq = p.Sq;
    const size_t qtok0 = (size_t)r * Sq;
    stage_head<SP>(Qs, p.Q + qtok0 * p.ldq + h * HD, p.ldq, Sq, tid);
    stage_head<SP>(Gs, p.dO + qtok0 * p.lddo + h * HD, p.lddo, Sq, tid);
    for (int i = tid; i < SP; i += ATT_THREADS) {
        traj_s[i] = (p.traj && i < S) ? p.traj[tok0 + i] : -1;
        kv_s[i] = (p.kvalid && i < S) ? p.kvalid[tok0 + i] : 1;
        lse_s[i] = i < Sq ? p.LSE[((size_t)r * p.H + h) * Sq + i] * (GENERIC ? 1.f : LOG2E) : INFINITY;  // +inf => P = 0 for padded queries
    }
    {   // D[q] = sum_d dO[q,d] * O[q,d]: 4 lanes per row (16 columns each), all rows' loads in flight at once
        constexpr int IT = SP * 4 / ATT_THREADS;
        float part[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int q = tid + i * ATT_THREADS;
            const int row = q >> 2, c = (q & 3) * 16;
            part[i] = 0.f;
            if (row < Sq) {
                const bf16_t* gp = p.dO + (qtok0 + row) * p.lddo + h * HD + c;
                const bf16_t* op = p.O + (qtok0 + row) * p.ldo + h * HD + c;
                part[i] = dot8(*(const bf16x8*)gp, *(const bf16x8*)op) + dot8(*(const bf16x8*)(gp + 8), *(const bf16x8*)(op + 8));
            }
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            float v = part[i];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            const int q = tid + i * ATT_THREADS;
            if ((q & 3) == 0) D_s[q >> 2] = v;
        }
    }
    __syncthreads();
    const unsigned char* kvp = p.kvalid ? kv_s : nullptr;
    const int ql = lane & 15, g = lane >> 4;
    const int ntile = (S + 15) / 16;
    const float sl2 = p.scale * LOG2E;
    const RowBase Qrow = att_row_base(Qs, lane), Grow = att_row_base(Gs, lane);
    const TrBase Qtr = att_tr_base(Qs, lane), Gtr = att_tr_base(Gs, lane);
#pragma unroll
    for (int t = 0; t < MAXKT; ++t) {
        const int kt = wid + 4 * t;
        if (kt >= ntile) break;
        const int keyl = kt * 16 + ql;  // this lane's key as the B-operand column
        const bool kok = keyl < S;
        if (t + 1 < MAXKT && kt + 4 < ntile) load_kv(t + 1, kvbuf[(t + 1) & 1]);
        const bf16x8 kf0 = kvbuf[t & 1][0], kf1 = kvbuf[t & 1][1], vf0 = kvbuf[t & 1][2], vf1 = kvbuf[t & 1][3];
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int w = 0; w < NKT / 2; ++w) {
            if (w * 32 < Sq) {
                float pv[8], dsv[8];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int qt = 2 * w + e2;
                    f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    s = mfma16(lds_row8i(Qrow.lo, qt * 16), kf0, s);
                    s = mfma16(lds_row8i(Qrow.hi, qt * 16), kf1, s);
                    dp = mfma16(lds_row8i(Grow.lo, qt * 16), vf0, dp);
                    dp = mfma16(lds_row8i(Grow.hi, qt * 16), vf1, dp);
                    // s[e]: query qt*16 + 4g + e, key keyl
                    if constexpr (GENERIC) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int q = qt * 16 + 4 * g + e;
                            float sv = s[e] * p.scale;
                            if (p.bias && q < S && kok) sv += p.bias[((size_t)h * S + q) * S + keyl];
                            const bool mk = (q >= Sq) || masked(p, q < Sq ? q : 0, keyl, traj_s, kvp);
                            const float pr = mk ? 0.f : __expf(sv - lse_s[q]);
                            const bool kp = !p.drop.thr || att_keep1(p.drop, att_drop_row(p, r, h, q < Sq ? q : 0) + keyl);
                            pv[e2 * 4 + e] = kp ? pr * p.drop.scale : 0.f;
                            dsv[e2 * 4 + e] = pr * ((kp ? dp[e] * p.drop.scale : 0.f) - D_s[q]) * p.scale;
                        }
                    } else {
                        // lse_s holds lse*log2e here (+inf for padded queries => P = 0); padded key columns are never stored
                        const f32x4 l4 = *(const f32x4*)(lse_s + 4 * g + qt * 16), d4 = *(const f32x4*)(D_s + 4 * g + qt * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float pr = __builtin_amdgcn_exp2f(s[e] * sl2 - l4[e]);
                            const int q = qt * 16 + 4 * g + e;
                            const bool kp = !p.drop.thr || att_keep1(p.drop, att_drop_row(p, r, h, q < Sq ? q : 0) + keyl);
                            pv[e2 * 4 + e] = kp ? pr * p.drop.scale : 0.f;
                            dsv[e2 * 4 + e] = pr * ((kp ? dp[e] * p.drop.scale : 0.f) - d4[e]) * p.scale;
                        }
                    }
                }
                const bf16x8 pa = pack8(pv), da = pack8(dsv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = mfma16(lds_tr8i(Gtr.d[dt], 32 * w, 32 * w + 16), pa, dv[dt]);   // dV^T / dK^T: rows = head dims, cols = keys
                    dk[dt] = mfma16(lds_tr8i(Qtr.d[dt], 32 * w, 32 * w + 16), da, dk[dt]);
                }
            }
        }
        if (kok) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x2 wk = {pack_bf2(dk[dt][0], dk[dt][1]), pack_bf2(dk[dt][2], dk[dt][3])};
                const u32x2 wv = {pack_bf2(dv[dt][0], dv[dt][1]), pack_bf2(dv[dt][2], dv[dt][3])};
                *(u32x2*)(p.dK + (tok0 + keyl) * p.ldd + h * HD + dt * 16 + 4 * g) = wk;
                *(u32x2*)(p.dV + (tok0 + keyl) * p.ldd + h * HD + dt * 16 + 4 * g) = wv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ exact-tile backward
// The fusion-encoder case (no mask / bias, (NKT-1)*16 < S <= NKT*16): every per-tile runtime predicate of the kernels above
// disappears (they cost SGPR spills through v_readlane/v_writelane and hundreds of register moves around the partially
// unrolled loops), the key / query pair loops are rolled, the softmax algebra is vector code (packed fp32), and
// D = rowsum(dO * O) is computed once (dQ kernel) and handed to the dK/dV kernel through a workspace instead of re-reading O.
template <int NKT>
__global__ void __launch_bounds__(ATT_THREADS, NKT <= 12 ? 3 : 2) attn_bwd_dq_exact_kernel(AttnArgs p) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 16, MAXQT = (NKT + 3) / 4;
    bf16_t* Ks = (bf16_t*)smem;
    bf16_t* Vs = Ks + SP * LDSROW;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const size_t tok0 = (size_t)r * p.S;
    const int S = p.S, Sq = p.Sq;
    stage_head<SP>(Ks, p.K + tok0 * p.ld + h * HD, p.ld, S, tid);
    stage_head<SP>(Vs, p.V + tok0 * p.ld + h * HD, p.ld, S, tid);
    const int ql = lane & 15, g = lane >> 4;
    const size_t qtok0 = (size_t)r * Sq;
    bf16x8 qall[MAXQT][4];
    float dall[MAXQT], lall[MAXQT];
#pragma unroll
    for (int t = 0; t < MAXQT; ++t) {
        const int q = (wid + 4 * t) * 16 + ql;
        const bool qok = q < Sq;
        const bf16_t* qp = p.Q + (qtok0 + (qok ? q : 0)) * p.ldq + h * HD + 8 * g;
        const bf16_t* gp = p.dO + (qtok0 + (qok ? q : 0)) * p.lddo + h * HD + 8 * g;
        const bf16_t* op = p.O + (qtok0 + (qok ? q : 0)) * p.ldo + h * HD + 8 * g;
        qall[t][0] = gld8(qp, qok); qall[t][1] = gld8(qp + 32, qok);
        qall[t][2] = gld8(gp, qok); qall[t][3] = gld8(gp + 32, qok);
        dall[t] = dot8(qall[t][2], gld8(op, qok)) + dot8(qall[t][3], gld8(op + 32, qok));
        lall[t] = qok ? p.LSE[((size_t)r * p.H + h) * Sq + q] * LOG2E : INFINITY;     // +inf => P = 0 for padded queries
    }
    __syncthreads();
    const RowBase Krow = att_row_base(Ks, lane), Vrow = att_row_base(Vs, lane);
    const TrBase Ktr = att_tr_base(Ks, lane);
    const int ntile = (Sq + 15) / 16;
    const float sl2 = p.scale * LOG2E;
    const f32x4 sl4 = {sl2, sl2, sl2, sl2}, sc4 = {p.scale, p.scale, p.scale, p.scale};
#pragma unroll
    for (int t = 0; t < MAXQT; ++t) {
        const int qt = wid + 4 * t;
        if (qt < ntile) {                                   // wave-uniform
            const int q = qt * 16 + ql;
            const bool qok = q < Sq;
            float D_q = dall[t];
            D_q += __shfl_xor(D_q, 16, 64);
            D_q += __shfl_xor(D_q, 32, 64);
            if (p.Dws && g == 0 && qok) p.Dws[((size_t)r * p.H + h) * Sq + q] = D_q;
            const f32x4 nl4 = {-lall[t], -lall[t], -lall[t], -lall[t]}, nd4 = {-D_q, -D_q, -D_q, -D_q};
            const unsigned long long drow = att_drop_row(p, r, h, qok ? q : 0);
            f32x4 dq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // padded keys need no mask: their K rows are zero, so their dS never reaches dQ
#pragma unroll 2
            for (int u = 0; u < NKT / 2; ++u) {
                const int off = u * 32 * LDSROW;
                float dsv[8];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int o2 = off + e2 * 16 * LDSROW;
                    f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sv = mfma16(*(const bf16x8*)(Krow.lo + o2), qall[t][0], sv);
                    sv = mfma16(*(const bf16x8*)(Krow.hi + o2), qall[t][1], sv);
                    dp = mfma16(*(const bf16x8*)(Vrow.lo + o2), qall[t][2], dp);
                    dp = mfma16(*(const bf16x8*)(Vrow.hi + o2), qall[t][3], dp);
                    const f32x4 x = sv * sl4 + nl4;
                    f32x4 pr;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pr[e] = __builtin_amdgcn_exp2f(x[e]);
                    if (p.drop.thr) {      // wave-uniform: dP = keep/(1-p) * (dO V^T), the forward's keep-mask regenerated
                        const unsigned keep = drop_keep4(p.drop, drow + (2 * u + e2) * 16 + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dp[e] = ((keep >> e) & 1u) ? dp[e] * p.drop.scale : 0.f;
                    }
                    const f32x4 ds = pr * (dp + nd4) * sc4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) dsv[e2 * 4 + e] = ds[e];
                }
                const bf16x8 da = pack8(dsv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    dq[dt] = mfma16(lds_tr8i(Ktr.d[dt] + off, 0, 16), da, dq[dt]);   // dQ^T: rows = head dims, cols = queries
            }
            if (qok) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const u32x2 w = {pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3])};
                    *(u32x2*)(p.dQ + (qtok0 + q) * p.lddq + h * HD + dt * 16 + 4 * g) = w;
                }
            }
        }
    }
}

template <int NKT>
__global__ void __launch_bounds__(ATT_THREADS, NKT <= 12 ? 3 : 2) attn_bwd_dkv_exact_kernel(AttnArgs p) {
    p.drop = drop_resolve(p.drop);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 16, MAXKT = (NKT + 3) / 4;
    bf16_t* Qs = (bf16_t*)smem;
    bf16_t* Gs = Qs + SP * LDSROW;  // dO
    float* lse_s = (float*)(Gs + SP * LDSROW);
    float* D_s = lse_s + SP;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const size_t tok0 = (size_t)r * p.S;
    const int S = p.S, Sq = p.Sq;
    bf16x8 kvbuf[2][4];
    auto load_kv = [&](int t, bf16x8 (&dst)[4]) {
        const int keyl = (wid + 4 * t) * 16 + (lane & 15);
        const bool kok = keyl < S;
        const bf16_t* kp = p.K + (tok0 + (kok ? keyl : 0)) * p.ld + h * HD + 8 * (lane >> 4);
        const bf16_t* vp = p.V + (tok0 + (kok ? keyl : 0)) * p.ld + h * HD + 8 * (lane >> 4);
        dst[0] = gld8(kp, kok); dst[1] = gld8(kp + 32, kok);
        dst[2] = gld8(vp, kok); dst[3] = gld8(vp + 32, kok);
    };
    load_kv(0, kvbuf[0]);
    const size_t qtok0 = (size_t)r * Sq;
    stage_head<SP>(Qs, p.Q + qtok0 * p.ldq + h * HD, p.ldq, Sq, tid);
    stage_head<SP>(Gs, p.dO + qtok0 * p.lddo + h * HD, p.lddo, Sq, tid);
    for (int i = tid; i < SP; i += ATT_THREADS) {
        lse_s[i] = i < Sq ? p.LSE[((size_t)r * p.H + h) * Sq + i] * LOG2E : INFINITY;   // +inf => P = 0 for padded queries
        D_s[i] = i < Sq ? p.Dws[((size_t)r * p.H + h) * Sq + i] : 0.f;
    }
    __syncthreads();
    const int ql = lane & 15, g = lane >> 4;
    const float sl2 = p.scale * LOG2E;
    const f32x4 sl4 = {sl2, sl2, sl2, sl2}, sc4 = {p.scale, p.scale, p.scale, p.scale};
    const RowBase Qrow = att_row_base(Qs, lane), Grow = att_row_base(Gs, lane);
    const TrBase Qtr = att_tr_base(Qs, lane), Gtr = att_tr_base(Gs, lane);
    const int nw = (Sq + 31) / 32;                        // query-tile pairs holding any real query
    const unsigned long long drow0 = att_drop_row(p, r, h, 0);   // dropout element index of (query 0, key 0) of this (row, head)
    const int S4 = (S + 3) & ~3;
#pragma unroll
    for (int t = 0; t < MAXKT; ++t) {
        const int kt = wid + 4 * t;
        if (kt < NKT) {                                     // all NKT key tiles are live (exact-tile launch)
            const int keyl = kt * 16 + ql;
            const bool kok = keyl < S;
            if (t + 1 < MAXKT && kt + 4 < NKT) load_kv(t + 1, kvbuf[(t + 1) & 1]);
            const bf16x8 kf0 = kvbuf[t & 1][0], kf1 = kvbuf[t & 1][1], vf0 = kvbuf[t & 1][2], vf1 = kvbuf[t & 1][3];
            f32x4 dk[4], dv[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 2
            for (int w = 0; w < nw; ++w) {
                const int off = w * 32 * LDSROW;
                float pv[8], dsv[8];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const int o2 = off + e2 * 16 * LDSROW;
                    f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    sv = mfma16(*(const bf16x8*)(Qrow.lo + o2), kf0, sv);
                    sv = mfma16(*(const bf16x8*)(Qrow.hi + o2), kf1, sv);
                    dp = mfma16(*(const bf16x8*)(Grow.lo + o2), vf0, dp);
                    dp = mfma16(*(const bf16x8*)(Grow.hi + o2), vf1, dp);
                    // sv[e]: query (2w+e2)*16 + 4g + e, key keyl
                    const f32x4 l4 = *(const f32x4*)(lse_s + w * 32 + e2 * 16 + 4 * g), d4 = *(const f32x4*)(D_s + w * 32 + e2 * 16 + 4 * g);
                    const f32x4 x = sv * sl4 - l4;
                    f32x4 pr, pd;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pr[e] = __builtin_amdgcn_exp2f(x[e]);
                    pd = pr;
                    if (p.drop.thr) {      // wave-uniform.  This layout holds 4 consecutive QUERIES of one key: one RNG word each
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int q = w * 32 + e2 * 16 + 4 * g + e;
                            const bool kp = att_keep1(p.drop, drow0 + (unsigned)((q < Sq ? q : 0) * S4 + keyl));
                            pd[e] = kp ? pr[e] * p.drop.scale : 0.f;       // dropped-out probabilities feed dV
                            dp[e] = kp ? dp[e] * p.drop.scale : 0.f;       // and dP = keep/(1-p) * (dO V^T)
                        }
                    }
                    const f32x4 ds = pr * (dp - d4) * sc4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pv[e2 * 4 + e] = pd[e]; dsv[e2 * 4 + e] = ds[e]; }
                }
                const bf16x8 pa = pack8(pv), da = pack8(dsv);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dv[dt] = mfma16(lds_tr8i(Gtr.d[dt] + off, 0, 16), pa, dv[dt]);   // dV^T / dK^T: rows = head dims, cols = keys
                    dk[dt] = mfma16(lds_tr8i(Qtr.d[dt] + off, 0, 16), da, dk[dt]);
                }
            }
            if (kok) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const u32x2 wk = {pack_bf2(dk[dt][0], dk[dt][1]), pack_bf2(dk[dt][2], dk[dt][3])};
                    const u32x2 wv = {pack_bf2(dv[dt][0], dv[dt][1]), pack_bf2(dv[dt][2], dv[dt][3])};
                    *(u32x2*)(p.dK + (tok0 + keyl) * p.ldd + h * HD + dt * 16 + 4 * g) = wk;
                    *(u32x2*)(p.dV + (tok0 + keyl) * p.ldd + h * HD + dt * 16 + 4 * g) = wv;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ single-pass backward
// The two exact-tile kernels above in one workgroup, so that Q, K, V, dO and O leave HBM once (8 instead of 12 [S,64] head slices
// of traffic per (row, head)).  LDS still holds only two operands at a time:
//   stage Q, dO (+ D = rowsum(dO * O) from the staging registers)  ->  dK, dV with this wave's K/V tiles streamed as fragments
//   ->  this wave's Q / dO query-tile fragments move LDS -> registers  ->  the K, V fragments are written over Q, dO  ->  dQ.
// VALU is the busiest unit of these kernels (PMC: 38 % VALU, 17 % MFMA, 23 % LDS, not overlapping), so the softmax algebra is
// two packed FMAs per element pair (-lse*log2e and -D*scale are what LDS holds) and the dropout hash is evaluated incrementally:
// the element-pair index P = P0 + q*(S4/2) + (key >> 1) is linear in (q, key), so lo(P)*C1 = lane constant + wave-uniform term.
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }

template <int NKT, bool DROP>
__global__ void __launch_bounds__(ATT_THREADS, NKT <= 12 ? 3 : 2) attn_bwd_fused_exact_kernel(AttnArgs p) {
    if (DROP) p.drop = drop_resolve(p.drop);
#ifdef ATT_TIMING
    const long long tm0 = __builtin_readcyclecounter();
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NKT * 16, MAXT = (NKT + 3) / 4, IT = SP * 8 / ATT_THREADS;
    constexpr bool EARLY_TR = ATT_EARLY_TR && NKT <= 12;
    constexpr bool KEEP_KV = true;        // K/V fragments of all of this wave's key tiles stay in registers and are written to LDS for
                                          // the dQ phase (no second read); NKT = 16 has no registers for that and re-reads K, V
    bf16_t* As = (bf16_t*)smem;          // Q, then K
    bf16_t* Bs = As + SP * LDSROW;       // dO, then V
    float* nl_s = (float*)(Bs + SP * LDSROW);   // -lse * log2(e)   (-inf for padded queries: P = 0)
    float* nd_s = nl_s + SP;                    // -D * scale
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    int r = blockIdx.x / p.H, h = blockIdx.x % p.H;
    att_xcd_item(p, blockIdx.x, gridDim.x, r, h);
    const size_t tok0 = (size_t)r * p.S, qtok0 = (size_t)r * p.Sq;
    const int S = p.S, Sq = p.Sq;
    const int ql = lane & 15, g = lane >> 4;
    constexpr int NBUF = KEEP_KV ? MAXT : 2;
    bf16x8 kvbuf[NBUF][4];
    auto load_kv = [&](int t, bf16x8 (&dst)[4]) {
        const int keyl = (wid + 4 * t) * 16 + ql;
        const bool kok = keyl < S;
        const bf16_t* kp = p.K + (tok0 + (kok ? keyl : 0)) * p.ld + h * HD + 8 * g;
        const bf16_t* vp = p.V + (tok0 + (kok ? keyl : 0)) * p.ld + h * HD + 8 * g;
        dst[0] = gld8(kp, kok); dst[1] = gld8(kp + 32, kok);
        dst[2] = gld8(vp, kok); dst[3] = gld8(vp + 32, kok);
    };
    load_kv(0, kvbuf[0]);
    if (KEEP_KV) {
#pragma unroll
        for (int t = 1; t < MAXT; ++t) if (wid + 4 * t < NKT) load_kv(t, kvbuf[t]);
    }
    stage_head<SP>(As, p.Q + qtok0 * p.ldq + h * HD, p.ldq, Sq, tid);
    {   // dO -> LDS, and D[row] = sum over the row's eight 16-byte chunks of dO . O (eight consecutive lanes hold one row)
        u32x4 w[IT];
        float part[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int q = tid + i * ATT_THREADS, row = q >> 3, c = q & 7;
            const bool ok = row < Sq;
            const bf16x8 gv = gld8(p.dO + (qtok0 + (ok ? row : 0)) * p.lddo + h * HD + c * 8, ok);
            const bf16x8 ov = gld8(p.O + (qtok0 + (ok ? row : 0)) * p.ldo + h * HD + c * 8, ok);
            w[i] = __builtin_bit_cast(u32x4, gv);
            part[i] = dot8(gv, ov);
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int q = tid + i * ATT_THREADS, row = q >> 3;
            *(u32x4*)(Bs + row * LDSROW + (((q & 7) ^ att_swz(row)) << 3)) = w[i];
            float d = part[i];
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            if ((q & 7) == 0) nd_s[row] = -d * p.scale;
        }
    }
    for (int i = tid; i < SP; i += ATT_THREADS)
        nl_s[i] = i < Sq ? -p.LSE[((size_t)r * p.H + h) * Sq + i] * LOG2E : -INFINITY;
    __syncthreads();
#ifdef ATT_TIMING
    const long long tm1 = __builtin_readcyclecounter();
#endif
    const float sl2 = p.scale * LOG2E;
    const float dsc = DROP ? p.drop.scale : 1.f, scd = p.scale * dsc;
    const f32x4 sl4 = {sl2, sl2, sl2, sl2}, scd4 = {scd, scd, scd, scd}, dsc4 = {dsc, dsc, dsc, dsc};
    const RowBase Arow = att_row_base(As, lane), Brow = att_row_base(Bs, lane);
    const TrBase Atr = att_tr_base(As, lane), Btr = att_tr_base(Bs, lane);
    const int S4 = (S + 3) & ~3, hS = S4 >> 1;
    AttDrop dh{};
    if (DROP) dh = att_drop_head(p, r, h);
    const unsigned thr = p.drop.thr;
    // ---- dK, dV: waves own key tiles; As = Q, Bs = dO
    {
        const int nw = (Sq + 31) / 32;                        // query-tile pairs holding any real query
        // dropout, this layout: a lane holds 4 consecutive QUERIES (4g + e) of one key (kt*16 + ql); the lanes ql, ql ^ 1 share their
        // RNG words (keys 2k, 2k+1 of one query), so the even lane hashes queries e = 0,1, the odd lane e = 2,3 and they swap halves
        const int odd = ql & 1;
        const unsigned dl_a = (unsigned)((4 * g + 2 * odd) * hS + (ql >> 1));   // lane part of delta for this lane's own two words
        const unsigned al_a = dh.a0 + dl_a * 0x9E3779B1u;
        const unsigned hSC = (unsigned)hS * 0x9E3779B1u;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int kt = wid + 4 * t;
            __builtin_amdgcn_sched_barrier(0);     // keep the unrolled key tiles apart (their loads would otherwise all be hoisted)
            if (kt < NKT) {
                const int keyl = kt * 16 + ql;
                const bool kok = keyl < S;
                if (!KEEP_KV && t + 1 < MAXT && kt + 4 < NKT) load_kv(t + 1, kvbuf[(t + 1) % NBUF]);
                const bf16x8 kf0 = kvbuf[t % NBUF][0], kf1 = kvbuf[t % NBUF][1], vf0 = kvbuf[t % NBUF][2], vf1 = kvbuf[t % NBUF][3];
                f32x4 dk[4], dv[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll (NKT <= 12 ? 2 : 1)       // full unroll (address bumps become immediates): measured 4 % slower
                for (int w = 0; w < nw; ++w) {
                    const int off = w * 32 * LDSROW;
                    float pv[8], dsv[8];
                    bf16x8 gfr[4], qfr[4];       // dO^T / Q^T fragments of this query pair: read first, their latency sits under the softmax block
                    if (EARLY_TR) {
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) { gfr[dt] = lds_tr8i(Btr.d[dt] + off, 0, 16); qfr[dt] = lds_tr8i(Atr.d[dt] + off, 0, 16); }
                    }
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const int o2 = off + e2 * 16 * LDSROW;
                        f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                        sv = mfma16(*(const bf16x8*)(Arow.lo + o2), kf0, sv);
                        sv = mfma16(*(const bf16x8*)(Arow.hi + o2), kf1, sv);
                        dp = mfma16(*(const bf16x8*)(Brow.lo + o2), vf0, dp);
                        dp = mfma16(*(const bf16x8*)(Brow.hi + o2), vf1, dp);
                        // sv[e]: query (2w+e2)*16 + 4g + e, key keyl
                        const f32x4 nl4 = *(const f32x4*)(nl_s + w * 32 + e2 * 16 + 4 * g), nd4 = *(const f32x4*)(nd_s + w * 32 + e2 * 16 + 4 * g);
                        const f32x4 x = fma4(sv, sl4, nl4);
                        f32x4 pr;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pr[e] = __builtin_amdgcn_exp2f(x[e]);
                        f32x4 pd = pr, tt = fma4(dp, scd4, nd4);      // tt = scale * (dP - D), dP = keep/(1-p) * (dO V^T)
                        if (DROP) {
                            pd = pr * dsc4;
                            unsigned keep;       // bit e: probability (query 4g + e, key keyl) kept
                            if (!dh.wrap) {
                                const int qb = w * 32 + e2 * 16;
                                const unsigned ua = (unsigned)(qb * hS + kt * 8) * 0x9E3779B1u;         // wave-uniform
                                const unsigned x0 = drop_mix((al_a + ua) ^ dh.hb), x1 = drop_mix((al_a + ua + hSC) ^ dh.hb);
                                // own half of my two words, and the partner's half of them
                                const unsigned mine = odd ? ((x0 >> 16) | (x1 & 0xffff0000u)) : ((x0 & 0xffffu) | (x1 << 16));
                                const unsigned give = odd ? ((x0 & 0xffffu) | (x1 << 16)) : ((x0 >> 16) | (x1 & 0xffff0000u));
                                const unsigned got = (unsigned)__builtin_amdgcn_mov_dpp((int)give, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
                                const unsigned q01 = odd ? got : mine, q23 = odd ? mine : got;      // 16-bit randoms of queries e = 0,1 | 2,3
                                keep = ((q01 & 0xffffu) >= thr ? 1u : 0u) | ((q01 >> 16) >= thr ? 2u : 0u) | ((q23 & 0xffffu) >= thr ? 4u : 0u) | ((q23 >> 16) >= thr ? 8u : 0u);
                            } else {
                                keep = 0;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int q = w * 32 + e2 * 16 + 4 * g + e;
                                    keep |= att_keep1(p.drop, (dh.p0 << 1) + (unsigned)(q * S4 + keyl)) ? (1u << e) : 0u;
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const bool kp = (keep >> e) & 1u;
                                pd[e] = kp ? pd[e] : 0.f;             // dropped-out probabilities feed dV
                                tt[e] = kp ? tt[e] : nd4[e];
                            }
                        }
                        const f32x4 ds = pr * tt;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pv[e2 * 4 + e] = pd[e]; dsv[e2 * 4 + e] = ds[e]; }
                    }
                    const bf16x8 pa = pack8(pv), da = pack8(dsv);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        if (!EARLY_TR) { gfr[dt] = lds_tr8i(Btr.d[dt] + off, 0, 16); qfr[dt] = lds_tr8i(Atr.d[dt] + off, 0, 16); }
                        dv[dt] = mfma16(gfr[dt], pa, dv[dt]);   // dV^T / dK^T: rows = head dims, cols = keys
                        dk[dt] = mfma16(qfr[dt], da, dk[dt]);
                    }
                }
                if (kok) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const u32x2 wk = {pack_bf2(dk[dt][0], dk[dt][1]), pack_bf2(dk[dt][2], dk[dt][3])};
                        const u32x2 wv = {pack_bf2(dv[dt][0], dv[dt][1]), pack_bf2(dv[dt][2], dv[dt][3])};
                        *(u32x2*)(p.dK + (tok0 + keyl) * p.ldd + h * HD + dt * 16 + 4 * g) = wk;
                        *(u32x2*)(p.dV + (tok0 + keyl) * p.ldd + h * HD + dt * 16 + 4 * g) = wv;
                    }
                }
            }
        }
    }
    // ---- this wave's query tiles: Q / dO row fragments, -lse and -D*scale move to registers before K, V replace Q, dO in LDS
    __builtin_amdgcn_sched_barrier(0);
#ifdef ATT_TIMING
    const long long tm2 = __builtin_readcyclecounter();
#endif
    bf16x8 qall[MAXT][4];
    float dall[MAXT], lall[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int qt = wid + 4 * t;
        if (qt < NKT) {
            const int o = qt * 16 * LDSROW;
            qall[t][0] = *(const bf16x8*)(Arow.lo + o); qall[t][1] = *(const bf16x8*)(Arow.hi + o);
            qall[t][2] = *(const bf16x8*)(Brow.lo + o); qall[t][3] = *(const bf16x8*)(Brow.hi + o);
            lall[t] = nl_s[qt * 16 + ql]; dall[t] = nd_s[qt * 16 + ql];
        }
    }
    __syncthreads();
    if (KEEP_KV) {
        // row fragment (key row kt*16 + ql, logical chunks g and g + 4) back to its swizzled LDS place; padded keys hold zeros
        const int f = att_swz(ql);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int kt = wid + 4 * t;
            if (kt < NKT) {
                bf16_t* ka = As + (kt * 16 + ql) * LDSROW;
                bf16_t* va = Bs + (kt * 16 + ql) * LDSROW;
                *(bf16x8*)(ka + ((g ^ f) << 3)) = kvbuf[t % NBUF][0]; *(bf16x8*)(ka + (((g + 4) ^ f) << 3)) = kvbuf[t % NBUF][1];
                *(bf16x8*)(va + ((g ^ f) << 3)) = kvbuf[t % NBUF][2]; *(bf16x8*)(va + (((g + 4) ^ f) << 3)) = kvbuf[t % NBUF][3];
            }
        }
    } else {
        stage_head<SP>(As, p.K + tok0 * p.ld + h * HD, p.ld, S, tid);
        stage_head<SP>(Bs, p.V + tok0 * p.ld + h * HD, p.ld, S, tid);
    }
    __syncthreads();
#ifdef ATT_TIMING
    const long long tm3 = __builtin_readcyclecounter();
#endif
    // ---- dQ: waves own query tiles; As = K, Bs = V
    {
        const int ntile = (Sq + 15) / 16;
        // dropout, this layout: a lane holds 4 consecutive KEYS (tile*16 + 4g + e) of one query: two RNG words
        const unsigned dl_b = (unsigned)(ql * hS + 2 * g);
        const unsigned al_b = dh.a0 + dl_b * 0x9E3779B1u;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int qt = wid + 4 * t;
            __builtin_amdgcn_sched_barrier(0);
            if (qt < ntile) {                                   // wave-uniform
                const int q = qt * 16 + ql;
                const bool qok = q < Sq;
                const f32x4 nl4 = {lall[t], lall[t], lall[t], lall[t]}, nd4 = {dall[t], dall[t], dall[t], dall[t]};
                f32x4 dq[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                // padded keys need no mask: their K rows are zero, so their dS never reaches dQ
#pragma unroll (NKT <= 12 ? 2 : 1)
                for (int u = 0; u < NKT / 2; ++u) {
                    const int off = u * 32 * LDSROW;
                    float dsv[8];
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const int o2 = off + e2 * 16 * LDSROW;
                        f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                        sv = mfma16(*(const bf16x8*)(Arow.lo + o2), qall[t][0], sv);
                        sv = mfma16(*(const bf16x8*)(Arow.hi + o2), qall[t][1], sv);
                        dp = mfma16(*(const bf16x8*)(Brow.lo + o2), qall[t][2], dp);
                        dp = mfma16(*(const bf16x8*)(Brow.hi + o2), qall[t][3], dp);
                        const f32x4 x = fma4(sv, sl4, nl4);
                        f32x4 pr;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pr[e] = __builtin_amdgcn_exp2f(x[e]);
                        f32x4 tt = fma4(dp, scd4, nd4);
                        if (DROP) {            // dP = keep/(1-p) * (dO V^T), the forward's keep-mask regenerated
                            unsigned keep;
                            if (!dh.wrap) {
                                const unsigned ub = (unsigned)(qt * 16 * hS + (2 * u + e2) * 8) * 0x9E3779B1u;   // wave-uniform
                                const unsigned r0 = drop_mix((al_b + ub) ^ dh.hb), r1 = drop_mix((al_b + ub + 0x9E3779B1u) ^ dh.hb);
                                keep = ((r0 & 0xffffu) >= thr ? 1u : 0u) | ((r0 >> 16) >= thr ? 2u : 0u) | ((r1 & 0xffffu) >= thr ? 4u : 0u) | ((r1 >> 16) >= thr ? 8u : 0u);
                            } else {
                                keep = drop_keep4(p.drop, (dh.p0 << 1) + (unsigned long long)(qok ? q : 0) * S4 + (2 * u + e2) * 16 + 4 * g);
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) tt[e] = ((keep >> e) & 1u) ? tt[e] : nd4[e];
                        }
                        const f32x4 ds = pr * tt;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dsv[e2 * 4 + e] = ds[e];
                    }
                    const bf16x8 da = pack8(dsv);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        dq[dt] = mfma16(lds_tr8i(Atr.d[dt] + off, 0, 16), da, dq[dt]);   // dQ^T: rows = head dims, cols = queries
                }
                if (qok) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const u32x2 w = {pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3])};
                        *(u32x2*)(p.dQ + (qtok0 + q) * p.lddq + h * HD + dt * 16 + 4 * g) = w;
                    }
                }
            }
        }
    }
#ifdef ATT_TIMING
    if (p.kvalid && lane == 0) {
        __builtin_amdgcn_sched_barrier(0);
        const long long tm4 = __builtin_readcyclecounter();
        unsigned* d = (unsigned*)p.kvalid + ((size_t)blockIdx.x * 4 + wid) * 4;
        d[0] = (unsigned)(tm1 - tm0); d[1] = (unsigned)(tm2 - tm1); d[2] = (unsigned)(tm3 - tm2); d[3] = (unsigned)(tm4 - tm3);
    }
#endif
}

template <int NKT>
static int launch_fwd(const AttnArgs& p, int rows, hipStream_t st) {
    const size_t lds = (size_t)2 * NKT * 16 * LDSROW * sizeof(bf16_t) + NKT * 16 * (sizeof(int) + 1);
    const bool generic = p.mask_mode != MASK_NONE || p.bias || p.kvalid;
    // persistent forward: S in (160, 192] (fusion encoder with short goals, S = 181) and S in (224, 256] (64-token instructions, S = 233)
    if constexpr (NKT == 12 || NKT == 16) {
        if (!generic && p.S > (NKT - 2) * 16) {
            const size_t ldsp = (size_t)2 * NKT * 16 * LDSROW * sizeof(bf16_t);
            static int slots = 0;
            if (!slots) {
                int dev = 0, n_cu = 0;
                HIP_CHECK_RET(hipGetDevice(&dev));
                HIP_CHECK_RET(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
                slots = 2 * n_cu;              // 2 workgroups per CU
            }
            const int nitems = rows * p.H;
            // (a tower group shares the two-workgroups-per-CU budget: the grouped launch carries one third of the slots per member)
            const int gslots = (slots / svla_group_size()) & ~7;
            return SVLA_LAUNCH_TWIN((attn_fwd_persist_kernel_body<NKT>), ATT_THREADS, 2, dim3(nitems < gslots ? nitems : gslots), dim3(ATT_THREADS), ldsp, st, p, nitems);
        }
    }
    if constexpr (NKT == 18) {           // S in (256, 288]: the ViT on 224 x 224 frames (16 x 16 patches + class token = 257): two workgroups per CU, exact tiles
        if (!generic && p.S > (NKT - 2) * 16 && p.Sq == p.S)
            return SVLA_LAUNCH_TWIN((attn_fwd_kernel_body<NKT, false, 4, true>), ATT_THREADS, 1, dim3(rows * p.H), dim3(ATT_THREADS), lds, st, p);
    }
    if constexpr (NKT >= 28) {           // one workgroup per CU (K + V > 80 KiB): eight waves share the LDS image
        if (!generic) {
            if (p.S > (NKT - 2) * 16 && p.Sq == p.S) return SVLA_LAUNCH_TWIN((attn_fwd_kernel_body<NKT, false, 8, true>), 512, 1, dim3(rows * p.H), dim3(512), lds, st, p);      // the ViT's 433 tokens
            return SVLA_LAUNCH_TWIN((attn_fwd_kernel_body<NKT, false, 8>), 512, 1, dim3(rows * p.H), dim3(512), lds, st, p);
        }
    }
    if (generic) return SVLA_LAUNCH_TWIN((attn_fwd_kernel_body<NKT, true>), ATT_THREADS, 1, dim3(rows * p.H), dim3(ATT_THREADS), lds, st, p);
    return SVLA_LAUNCH_TWIN((attn_fwd_kernel_body<NKT, false>), ATT_THREADS, 1, dim3(rows * p.H), dim3(ATT_THREADS), lds, st, p);
}
static int g_attn_no_decode = 0;      // svla_attn_bwd_two_pass(4): single-query forwards on the tile kernels (A/B, tests)
static int g_attn_bwd_two_pass = 0;   // svla_attn_bwd_two_pass(1): the dQ + dK/dV kernel pair instead of the single-pass kernel (A/B, tests)
static int g_attn_xcd_rows = 1;       // svla_attn_bwd_two_pass(2): single-pass kernel with row-major items (A/B of the XCD mapping)
extern "C" int svla_attn_bwd_two_pass(int on) { g_attn_bwd_two_pass = on & 1; g_attn_xcd_rows = !(on & 2); g_attn_no_decode = (on & 4) ? 1 : 0; return 0; }

template <int NKT>
static int launch_bwd(const AttnArgs& p, int rows, hipStream_t st) {
    const size_t lds_q = (size_t)2 * NKT * 16 * LDSROW * sizeof(bf16_t) + NKT * 16 * (sizeof(int) + 1);
    const size_t lds_kv = (size_t)2 * NKT * 16 * LDSROW * sizeof(bf16_t) + NKT * 16 * (2 * sizeof(float) + sizeof(int) + 1);
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<NKT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<NKT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv));
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<NKT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<NKT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv));
        attr = true;
    }
    const bool generic = p.mask_mode != MASK_NONE || p.bias || p.kvalid;
    // exact-tile backward: correct for any S <= NKT*16 (padded keys have zero K / V rows, padded queries lse = +inf); used where at most
    // three of the NKT key tiles are padding
    if constexpr (NKT <= 16) if (!generic && (p.Dws || !g_attn_bwd_two_pass) && p.S > (NKT - 3) * 16) {
        const size_t le_q = (size_t)2 * NKT * 16 * LDSROW * sizeof(bf16_t);
        const size_t le_kv = le_q + NKT * 16 * 2 * sizeof(float);
        static bool attr_e = false;
        if (!attr_e) {
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_dq_exact_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)le_q));
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_dkv_exact_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)le_kv));
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_fused_exact_kernel<NKT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)le_kv));
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_fused_exact_kernel<NKT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)le_kv));
            attr_e = true;
        }
        if (g_attn_bwd_two_pass) {
            hipLaunchKernelGGL((attn_bwd_dq_exact_kernel<NKT>), dim3(rows * p.H), dim3(ATT_THREADS), le_q, st, p);
            hipLaunchKernelGGL((attn_bwd_dkv_exact_kernel<NKT>), dim3(rows * p.H), dim3(ATT_THREADS), le_kv, st, p);
        } else {
#ifdef ATT_TIMING
            AttnArgs pt = p; if (getenv("SVLA_ATTN_DBGBUF")) pt.kvalid = (const unsigned char*)strtoull(getenv("SVLA_ATTN_DBGBUF"), nullptr, 16);
            if (p.drop.thr) { hipLaunchKernelGGL((attn_bwd_fused_exact_kernel<NKT, true>), dim3(rows * p.H), dim3(ATT_THREADS), le_kv, st, pt); return svla_launch_status(); }
#endif
            if (p.drop.thr) hipLaunchKernelGGL((attn_bwd_fused_exact_kernel<NKT, true>), dim3(rows * p.H), dim3(ATT_THREADS), le_kv, st, p);
            else hipLaunchKernelGGL((attn_bwd_fused_exact_kernel<NKT, false>), dim3(rows * p.H), dim3(ATT_THREADS), le_kv, st, p);
        }
        return svla_launch_status();
    }
    if (generic) {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<NKT, true>), dim3(rows * p.H), dim3(ATT_THREADS), lds_q, st, p);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<NKT, true>), dim3(rows * p.H), dim3(ATT_THREADS), lds_kv, st, p);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<NKT, false>), dim3(rows * p.H), dim3(ATT_THREADS), lds_q, st, p);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<NKT, false>), dim3(rows * p.H), dim3(ATT_THREADS), lds_kv, st, p);
    }
    return svla_launch_status();
}

// ------------------------------------------------------------------------------------------------ single-query ("decode") forward
// Sq == 1, no bias / trajectory mask / dropout: the KV-cached acting step of the llama decoder (one new token per env against a cache window of up to 500
// slots of which kvalid marks the env's current episode; allenact_dino_transformer.py:388-397) and the pruned last fusion layer in eval mode.  The tile kernels
// stage all S keys of K and V into LDS for that one query (128 KiB and 73 us per launch at S = 500, whatever the window); here a workgroup reads ONLY the valid keys,
// straight from global memory: thread (g = tid / 8, c = tid % 8) owns the 16-byte chunk c of keys g, g + 32, ...; scores are reduced over the eight lanes of a key,
// the softmax over the workgroup, P.V over the 32 key groups.  Same arithmetic as the MFMA path: bf16 products, fp32 accumulation, probabilities rounded to bf16
// before the product with V.
#define DEC_THREADS 256
#define DEC_MAXB 16          // S <= 512: 16 blocks of 32 keys
__device__ __forceinline__ void attn_decode_kernel_body(AttnArgs p) {
    __shared__ float red[4][2];
    __shared__ float accs[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = tid >> 3, c = tid & 7;
    const int r = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int S = p.S;
    const size_t tok0 = (size_t)r * p.kv_rows;
    const unsigned char* kvr = p.kvalid ? p.kvalid + (size_t)r * S : nullptr;
    const float sl2 = p.scale * LOG2E;
    float qv[8];
    {
        const bf16x8 q8 = *(const bf16x8*)(p.Q + (size_t)r * p.ldq + h * HD + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = bf2f((bf16_t)q8[e]);
    }
    const int nb = (S + 31) >> 5;
    float sc[DEC_MAXB];
    unsigned vmask = 0;
    float mx = -INFINITY;
#pragma unroll
    for (int b = 0; b < DEC_MAXB; ++b) {
        sc[b] = -INFINITY;
        if (b < nb) {
            const int key = b * 32 + g;
            const bool ok = key < S && (!kvr || kvr[key]);
            if (ok) {
                const bf16x8 k8 = *(const bf16x8*)(p.K + (tok0 + key) * p.ld + h * HD + c * 8);
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(qv[e], bf2f((bf16_t)k8[e]), d);
                sc[b] = d;
                vmask |= 1u << b;
            }
            // the eight lanes of a key hold its eight partial dot products (ok is uniform over them)
            float d = ok ? sc[b] : 0.f;
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            sc[b] = ok ? d * sl2 : -INFINITY;
            mx = fmaxf(mx, sc[b]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lane == 0) red[wid][0] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0], red[1][0]), fmaxf(red[2][0], red[3][0]));
    if (mx == -INFINITY) mx = 0.f;
    float acc[8], lsum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int b = 0; b < DEC_MAXB; ++b) {
        if (b < nb && ((vmask >> b) & 1u)) {
            const float pr = __builtin_amdgcn_exp2f(sc[b] - mx);
            lsum += pr;
            const float pb = bf2f(f2bf(pr));                         // the MFMA path's operand rounding
            const bf16x8 v8 = *(const bf16x8*)(p.V + (tok0 + b * 32 + g) * p.ld + h * HD + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pb, bf2f((bf16_t)v8[e]), acc[e]);
        }
    }
    // every key was counted by its eight lanes: sum over lanes / 8; the accumulators: over the lanes that share chunk c
    lsum += __shfl_xor(lsum, 1, 64); lsum += __shfl_xor(lsum, 2, 64); lsum += __shfl_xor(lsum, 4, 64);
    lsum += __shfl_xor(lsum, 8, 64); lsum += __shfl_xor(lsum, 16, 64); lsum += __shfl_xor(lsum, 32, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = acc[e];
        a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
        acc[e] = a;
    }
    if (lane == 0) red[wid][1] = lsum * 0.125f;
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) accs[wid][lane * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        const float ls = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        const float o = (accs[0][tid] + accs[1][tid]) + (accs[2][tid] + accs[3][tid]);
        const float inv = ls > 0.f ? 1.f / ls : 0.f;
        p.O[(size_t)r * p.ldo + h * HD + tid] = f2bf(o * inv);
        if (p.LSE && tid == 0) p.LSE[(size_t)r * p.H + h] = (mx + __log2f(ls)) * LN2;
    }
}
__global__ void __launch_bounds__(DEC_THREADS) attn_decode_kernel(AttnArgs p) { attn_decode_kernel_body(p); }

extern "C" int svla_attn_fwd_bf16(const bf16_t* Q, const bf16_t* K, const bf16_t* V, long ld, bf16_t* O, long ldo, float* LSE,
                                  int rows, int S, int H, int head_dim, float scale, int mask_mode, const int* traj,
                                  const float* bias, const unsigned char* kvalid, int Sq, long ldq, int kv_rows, const svla_dropout* drop,
                                  void* stream) {
    if (head_dim != HD || rows <= 0 || S <= 0 || S > 512 || (ld % 8) || H <= 0 || (kv_rows > 0 && kv_rows < S)) return SVLA_EINVAL;
    if (mask_mode == MASK_BLOCK_CAUSAL && !traj) return SVLA_EINVAL;
    if (Sq < 0 || Sq > S || (Sq > 0 && (ldq % 8))) return SVLA_EINVAL;
    AttnArgs p{};
    p.Q = Q; p.K = K; p.V = V; p.ld = ld; p.O = O; p.ldo = ldo; p.LSE = LSE; p.traj = traj; p.bias = bias; p.kvalid = kvalid;
    p.S = S; p.H = H; p.mask_mode = mask_mode; p.scale = scale;
    p.Sq = Sq > 0 ? Sq : S; p.ldq = Sq > 0 ? ldq : ld;
    p.kv_rows = kv_rows > 0 ? kv_rows : S;
    p.drop = drop_cfg(drop);
    p.xcd_rows = 0;      // (measured on the persistent forward: +1.5 % time with whole rows per XCD -- its next-item prefetch already hides the fetch; -1.4 % on the backward)
    hipStream_t st = (hipStream_t)stream;
    if (p.Sq == 1 && !bias && mask_mode == MASK_NONE && !p.drop.thr && !g_attn_no_decode) {      // one query per row: read the valid keys only (KV-cached acting step)
        SVLA_LAUNCH(attn_decode_kernel, attn_decode_kernel_body, DEC_THREADS, 1, dim3(rows * H), dim3(DEC_THREADS), 0, st, p);
        return svla_launch_status();
    }
    if (S <= 64) return launch_fwd<4>(p, rows, st);
    if (S <= 128) return launch_fwd<8>(p, rows, st);
    if (S <= 192) return launch_fwd<12>(p, rows, st);
    if (S <= 256) return launch_fwd<16>(p, rows, st);
    if (S <= 288) return launch_fwd<18>(p, rows, st);
    if (S <= 448) return launch_fwd<28>(p, rows, st);
    return launch_fwd<32>(p, rows, st);
}

extern "C" int svla_attn_bwd_bf16(const bf16_t* Q, const bf16_t* K, const bf16_t* V, long ld, const bf16_t* O, long ldo,
                                  const float* LSE, const bf16_t* dO, long lddo, bf16_t* dQ, bf16_t* dK, bf16_t* dV, long ldd,
                                  int rows, int S, int H, int head_dim, float scale, int mask_mode, const int* traj,
                                  const float* bias, const unsigned char* kvalid, int Sq, long ldq, long lddq, float* D_ws,
                                  const svla_dropout* drop, void* stream) {
    if (head_dim != HD || rows <= 0 || S <= 0 || S > 256 || (ld % 8) || (lddo % 8) || H <= 0) return SVLA_EINVAL;
    if (mask_mode == MASK_BLOCK_CAUSAL && !traj) return SVLA_EINVAL;
    if (Sq < 0 || Sq > S || (Sq > 0 && ((ldq % 8) || (lddq % 8)))) return SVLA_EINVAL;
    AttnArgs p{};
    p.Sq = Sq > 0 ? Sq : S; p.ldq = Sq > 0 ? ldq : ld; p.lddq = Sq > 0 ? lddq : ldd; p.kv_rows = S;
    p.Q = Q; p.K = K; p.V = V; p.ld = ld; p.O = (bf16_t*)O; p.ldo = ldo; p.LSE = (float*)LSE; p.dO = dO; p.lddo = lddo;
    p.dQ = dQ; p.dK = dK; p.dV = dV; p.ldd = ldd; p.traj = traj; p.bias = bias; p.kvalid = kvalid;
    p.S = S; p.H = H; p.mask_mode = mask_mode; p.scale = scale; p.Dws = D_ws; p.drop = drop_cfg(drop);
    p.xcd_rows = (g_attn_xcd_rows && H == 8) ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    if (S <= 64) return launch_bwd<4>(p, rows, st);
    if (S <= 128) return launch_bwd<8>(p, rows, st);
    if (S <= 192) return launch_bwd<12>(p, rows, st);
    return launch_bwd<16>(p, rows, st);
}
