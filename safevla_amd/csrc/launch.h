// Tower-grouped launches (round 6).
//
// The three towers of the actor-critic (actor, reward critic, cost critic: separate_actor_critic.py:27-37) execute the SAME kernel
// sequence on the same shapes with different weights.  At an acting step every one of those kernels is small (64 ... 11 584 rows): three
// streams of ~100 dependent launches each, where a chip-filling kernel of one tower holds up the other two (profiles/r05_policy_step_kernel_stats.txt).
// With a capture open (svla_group_begin), a launch site that goes through SVLA_LAUNCH does not launch: it stores the kernel's arguments.
// svla_group_end then issues, for launch j of the call, ONE grid whose blockIdx.z selects the member's argument block -- three times the
// workgroups per dispatch, one dependency chain instead of three.  Anything that is not identical across the members (kernel, grid, block,
// LDS bytes) falls back to one launch per member, in member order, on the same stream: always correct, only slower.
//
// A kernel takes part by being written as  __device__ body(args...)  +  __global__ kernel(args...) { body(args...); }  (the single-launch kernel keeps
// its name, signature and code); its grouped twin is svla_grouped<&body, ...>, generated here.
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>
#include <utility>

#define SVLA_MAXG 3             // towers
#define SVLA_MAXQ 8             // launches one C-ABI call may issue
#define SVLA_DEFER_ARG_BYTES 512

template <typename... Ts> struct ArgPack;
template <> struct ArgPack<> {};
template <typename T, typename... Ts> struct ArgPack<T, Ts...> { T head; ArgPack<Ts...> tail; };

template <size_t I, typename T, typename... Ts>
__host__ __device__ __forceinline__ const auto& pack_get(const ArgPack<T, Ts...>& p) {
    if constexpr (I == 0) return p.head;
    else return pack_get<I - 1>(p.tail);
}
template <typename T, typename... Ts> inline ArgPack<T, Ts...> pack_make(const T& h, const Ts&... t) {
    if constexpr (sizeof...(Ts) == 0) return ArgPack<T>{h, ArgPack<>{}};
    else return ArgPack<T, Ts...>{h, pack_make<Ts...>(t...)};
}

template <typename P> struct GroupedArgs { P a[SVLA_MAXG]; };

template <auto Body, typename... Ts, size_t... I>
__device__ __forceinline__ void svla_call_body(const ArgPack<Ts...>& p, std::index_sequence<I...>) { Body(pack_get<I>(p)...); }

// the grouped twin: blockIdx.z = member (the argument block is read from the kernarg segment at a wave-uniform offset)
template <auto Body, int MAXT, int MINB, typename... Ts>
__global__ void __launch_bounds__(MAXT, MINB) svla_grouped(GroupedArgs<ArgPack<Ts...>> g) {
    svla_call_body<Body, Ts...>(g.a[blockIdx.z], std::index_sequence_for<Ts...>{});
}

struct DeferredLaunch {
    // issues members m[0..n): grouped when n > 1 (the caller has checked that they are identical in everything but the arguments)
    int (*flush)(const DeferredLaunch* const* m, int n, hipStream_t s);
    dim3 grid, block;
    unsigned smem;
    hipStream_t stream;
    alignas(16) unsigned char args[SVLA_DEFER_ARG_BYTES];
};
struct GroupCapture {
    int size = 0, member = 0;
    int n[SVLA_MAXG] = {0, 0, 0};
    DeferredLaunch q[SVLA_MAXG][SVLA_MAXQ];
    int overflow = 0;
    long grouped = 0, single = 0;      // statistics since svla_group_stats was last read
};
GroupCapture* svla_group_capture();    // misc.hip: this thread's open capture, or nullptr
int svla_group_size();                 // members of the open capture (1: none) -- dispatchers size persistent grids / choose kernels for the GROUP's work

// issue n >= 1 members' argument blocks as ONE grid of the grouped twin (grid.z = n)
template <auto Body, int MAXT, int MINB, typename... Ts>
inline int svla_twin_launch(const DeferredLaunch* const* m, int n, hipStream_t s) {
    using Pack = ArgPack<Ts...>;
    GroupedArgs<Pack> g;
    for (int i = 0; i < SVLA_MAXG; ++i) memcpy((void*)&g.a[i], m[i < n ? i : 0]->args, sizeof(Pack));
    static unsigned attr_smem = 0;      // dynamic LDS this twin has been cleared for (the single-launch kernels set theirs at their launch sites)
    if (m[0]->smem > attr_smem) {
        const hipError_t e = hipFuncSetAttribute((const void*)svla_grouped<Body, MAXT, MINB, Ts...>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m[0]->smem);
        if (e != hipSuccess) return (int)e;
        attr_smem = m[0]->smem;
    }
    dim3 grid = m[0]->grid;
    grid.z = (unsigned)n;
    hipLaunchKernelGGL((svla_grouped<Body, MAXT, MINB, Ts...>), grid, m[0]->block, m[0]->smem, s, g);
    return (int)hipGetLastError();
}
template <typename... Ts> inline void svla_defer_fill(DeferredLaunch& d, int (*flush)(const DeferredLaunch* const*, int, hipStream_t), dim3 grid, dim3 block, size_t smem,
                                                      hipStream_t s, const Ts&... a) {
    static_assert(sizeof(ArgPack<Ts...>) <= SVLA_DEFER_ARG_BYTES, "deferred argument block too small");
    d.flush = flush;
    d.grid = grid; d.block = block; d.smem = (unsigned)smem; d.stream = s;
    const ArgPack<Ts...> p = pack_make<Ts...>(a...);
    memcpy(d.args, (const void*)&p, sizeof(p));
}

// A kernel with a same-named single-launch __global__ (Kern) and a grouped twin generated from its body
template <auto Kern, auto Body, int MAXT, int MINB> struct Launch;
template <typename... Ts, void (*Kern)(Ts...), void (*Body)(Ts...), int MAXT, int MINB>
struct Launch<Kern, Body, MAXT, MINB> {
    using Pack = ArgPack<Ts...>;
    template <size_t... I>
    static void single(const DeferredLaunch& d, hipStream_t s, std::index_sequence<I...>) {
        Pack p;
        memcpy((void*)&p, d.args, sizeof(Pack));
        hipLaunchKernelGGL(Kern, d.grid, d.block, d.smem, s, pack_get<I>(p)...);
    }
    static int flush(const DeferredLaunch* const* m, int n, hipStream_t s) {
        if (n == 1) {
            single(*m[0], s, std::index_sequence_for<Ts...>{});
            return (int)hipGetLastError();
        }
        return svla_twin_launch<Body, MAXT, MINB, Ts...>(m, n, s);
    }
    static void go(dim3 grid, dim3 block, size_t smem, hipStream_t s, Ts... a) {
        GroupCapture* gc = svla_group_capture();
        if (!gc || gc->n[gc->member] >= SVLA_MAXQ) {      // (no call issues SVLA_MAXQ launches; if one ever does, it is launched at once and the capture is marked)
            if (gc) gc->overflow = 1;
            hipLaunchKernelGGL(Kern, grid, block, smem, s, a...);
            return;
        }
        svla_defer_fill<Ts...>(gc->q[gc->member][gc->n[gc->member]], &flush, grid, block, smem, s, a...);
        gc->n[gc->member] += 1;
    }
};
// A kernel that exists ONLY as the grouped form: a single launch is a group of one (grid.z = 1).  For the attention forward kernels: their bodies inlined into a
// plain same-named __global__ compiled worse than the original kernels did (attn_fwd_persist_kernel<12>: 109 spilled VGPRs against 6, 2.99 -> 5.97 ms per launch
// in the update -- caught by an A/B against the round-5 tree on one box), while the same body inside svla_grouped<> compiles like the original
template <auto Body, int MAXT, int MINB> struct LaunchTwin;
template <typename... Ts, void (*Body)(Ts...), int MAXT, int MINB>
struct LaunchTwin<Body, MAXT, MINB> {
    static int flush(const DeferredLaunch* const* m, int n, hipStream_t s) { return svla_twin_launch<Body, MAXT, MINB, Ts...>(m, n, s); }
    static int go(dim3 grid, dim3 block, size_t smem, hipStream_t s, Ts... a) {
        GroupCapture* gc = svla_group_capture();
        if (!gc || gc->n[gc->member] >= SVLA_MAXQ) {
            if (gc) gc->overflow = 1;
            DeferredLaunch d;
            svla_defer_fill<Ts...>(d, &flush, grid, block, smem, s, a...);
            const DeferredLaunch* one = &d;
            return flush(&one, 1, s);
        }
        svla_defer_fill<Ts...>(gc->q[gc->member][gc->n[gc->member]], &flush, grid, block, smem, s, a...);
        gc->n[gc->member] += 1;
        return 0;
    }
};
// kern / body may be template-ids with commas: pass them in parentheses
#define SVLA_LAUNCH(kern, body, maxt, minb, grid, block, smem, stream, ...) \
    Launch<&kern, &body, maxt, minb>::go(grid, block, smem, stream, __VA_ARGS__)
#define SVLA_LAUNCH_TWIN(body, maxt, minb, grid, block, smem, stream, ...) \
    LaunchTwin<&body, maxt, minb>::go(grid, block, smem, stream, __VA_ARGS__)
