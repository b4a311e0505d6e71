"""Tower-grouped launches (csrc/launch.h, include/svla.h: svla_group_begin / svla_replay_calls_grouped; round 6).

The three towers (separate_actor_critic.py:27-37) run the same kernels on the same shapes with different weights; a grouped launch issues the
three as ONE grid (blockIdx.z = tower).  What must hold: a grouped launch is bit-identical to the three single launches (same kernels, same
per-member grids), members that differ in anything but their arguments fall back to single launches, and the acting step replayed as grouped
launches equals the three-stream replay step for step -- logits, values and the KV caches, with the train-mode dropout on."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF16 = torch.bfloat16


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def _grouped(fn_per_member, members=3):
    """run fn_per_member(m) for m in range(members) inside one launch-group capture on the current stream"""
    from safevla_amd import _lib

    L = _lib.lib()
    L.call("svla_group_begin", members)
    try:
        outs = []
        for m in range(members):
            L.call("svla_group_member", m)
            outs.append(fn_per_member(m))
    finally:
        rc = L.cdll.svla_group_end(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
    return outs


@pytest.mark.parametrize("M,N,K,flavour", [(64, 1536, 512, "plain"), (768, 512, 512, "residual_drop"), (11584, 1536, 512, "plain"),
                                           (11584, 2048, 512, "relu_drop"), (11584, 512, 2048, "residual_drop"), (333, 512, 512, "relu")])
def test_grouped_gemm_equals_single_launches(M, N, K, flavour):
    """svla_gemm_nt_bf16 of three 'towers' in one capture == the three calls alone, bit for bit (128-tile kernel and the persistent 256-tile kernel,
    dropout counters included), and the launches really were grouped."""
    _need_gpu()
    from safevla_amd import ops

    g = torch.Generator(device=DEV).manual_seed(M + N)
    A = [torch.randn(M, K, device=DEV, generator=g).to(BF16) for _ in range(3)]
    W = [(torch.randn(N, K, device=DEV, generator=g) * 0.05).to(BF16) for _ in range(3)]
    b = [torch.randn(N, device=DEV, generator=g) for _ in range(3)]
    R = [torch.randn(M, N, device=DEV, generator=g).to(BF16) for _ in range(3)]
    kw = {"plain": {}, "relu": dict(act=ops.ACT_RELU), "relu_drop": dict(act=ops.ACT_RELU), "residual_drop": {}}[flavour]

    def call(m):
        extra = dict(kw)
        if "drop" in flavour:
            extra["drop"] = ops.Dropout(1234 + m, 3, 0.1)
        if "residual" in flavour:
            extra["residual"] = R[m]
        return ops.gemm_nt(A[m], W[m], M, N, K, bias=b[m], **extra)

    want = [call(m) for m in range(3)]
    ops.group_stats()
    got = _grouped(call)
    torch.cuda.synchronize()
    grouped, single = ops.group_stats()
    assert grouped >= 1 and single == 0, (grouped, single)
    for m in range(3):
        assert torch.equal(got[m], want[m]), (m, float((got[m].float() - want[m].float()).abs().max()))
    assert not torch.equal(got[0], got[1])


def test_grouped_norm_attention_and_glue_kernels():
    """LayerNorm / RMSNorm forward, the T5-style masked attention, the decode attention, SwiGLU: grouped == single, bit for bit"""
    _need_gpu()
    from safevla_amd import ops

    g = torch.Generator(device=DEV).manual_seed(7)
    rows, D, H = 768, 512, 8
    x = [torch.randn(rows, D, device=DEV, generator=g).to(BF16) for _ in range(3)]
    gam = [torch.rand(D, device=DEV, generator=g) + 0.5 for _ in range(3)]
    bet = [torch.randn(D, device=DEV, generator=g) for _ in range(3)]
    norm = lambda m: ops.norm_fwd(x[m], gam[m], bet[m], 1e-5, rows, D=D)[0]
    rms = lambda m: ops.norm_fwd(x[m], gam[m], None, 1e-6, rows, rms=True, save_stats=False, D=D)[0]
    qkv = [torch.randn(64 * 12, 3 * D, device=DEV, generator=g).to(BF16) for _ in range(3)]
    bias = [torch.randn(H, 12, 12, device=DEV, generator=g) for _ in range(3)]
    att = lambda m: ops.attn_fwd(qkv[m], qkv[m][:, D:], qkv[m][:, 2 * D:], 3 * D, 64, 12, H, 1.0, bias=bias[m], save_lse=False, drop=ops.Dropout(5 + m, 1, 0.1))[0]
    ab = [torch.randn(64, 4096, device=DEV, generator=g).to(BF16) for _ in range(3)]
    swi = lambda m: ops.swiglu_fwd(ab[m], 64, 2048)
    for name, fn in (("layernorm", norm), ("rmsnorm", rms), ("t5 attention", att), ("swiglu", swi)):
        want = [fn(m) for m in range(3)]
        ops.group_stats()
        got = _grouped(fn)
        torch.cuda.synchronize()
        grouped, single = ops.group_stats()
        assert grouped >= 1 and single == 0, (name, grouped, single)
        for m in range(3):
            assert torch.equal(got[m], want[m]), (name, m)


def test_members_with_different_shapes_fall_back_to_single_launches():
    _need_gpu()
    from safevla_amd import ops

    g = torch.Generator(device=DEV).manual_seed(3)
    Ms = [64, 300, 64]          # 1 / 3 / 1 row tiles: the grids differ
    A = [torch.randn(M, 512, device=DEV, generator=g).to(BF16) for M in Ms]
    W = [(torch.randn(512, 512, device=DEV, generator=g) * 0.05).to(BF16) for _ in range(3)]
    call = lambda m: ops.gemm_nt(A[m], W[m], Ms[m], 512, 512)
    want = [call(m) for m in range(3)]
    ops.group_stats()
    got = _grouped(call)
    torch.cuda.synchronize()
    grouped, single = ops.group_stats()
    assert grouped == 0 and single == 3, (grouped, single)
    for m in range(3):
        assert torch.equal(got[m], want[m])


def test_capture_protocol_errors():
    _need_gpu()
    from safevla_amd import _lib

    L = _lib.lib().cdll
    assert L.svla_group_end(None) != 0                      # nothing open
    assert L.svla_group_begin(4) != 0 and L.svla_group_begin(0) != 0
    assert L.svla_group_begin(2) == 0
    assert L.svla_group_begin(2) != 0                       # captures do not nest
    assert L.svla_group_member(2) != 0 and L.svla_group_member(1) == 0
    assert L.svla_group_end(None) == 0                      # empty capture: nothing issued


@pytest.mark.parametrize("train", [True, False])
def test_acting_steps_grouped_equal_three_stream_replay(train):
    """The recorded acting step at 8 envs, stepped 12 times through the tower-grouped replay and through the three-stream replay from identical
    states: same logits / values / cost values at every step and the same KV caches at the end -- with dropout on (``train``: device-resident seeds
    advance identically) and off.  And the grouped path really grouped (no single-launch fall-back left in the chain)."""
    _need_gpu()
    from safevla_amd import ops
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    B, n = 8, 12
    outs = {}
    for mode in (True, False):
        torch.manual_seed(11)
        m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)
        m.train(train)
        st, _, _ = fill_synthetic_rollout(m, SynthSpec(T=n + 2, B=B, L=12, task="PickUp", seed=5), device=DEV)
        for t in m.towers:
            t.time_step_counter, t._kv = 0, None
        m.grouped_towers = mode
        m.enable_acting_plans(True)
        res = []
        ops.group_stats()
        with torch.no_grad():
            for t in range(n):
                o, _ = m({k: v[t:t + 1] for k, v in st.observations.items()}, None, st.prev_actions[t:t + 1], st.masks[t:t + 1])
                res.append((o.distributions.logits.float().clone(), o.values.float().clone(), o.c_values.float().clone()))
        torch.cuda.synchronize()
        stats = ops.group_stats()
        kv = [c.clone() for t in m.towers for c in t._kv]
        outs[mode] = (res, kv, stats)
        del m, st
    (rg, kvg, sg), (rs, kvs, ss) = outs[True], outs[False]
    assert sg[0] > 50 * (n - 1) and sg[1] == 0, sg              # ~100 grouped launches per step, none issued singly
    assert ss == (0, 0), ss
    for t in range(n):
        for a, b in zip(rg[t], rs[t]):
            assert torch.equal(a, b), (t, float((a - b).abs().max()))
    for a, b in zip(kvg, kvs):
        assert torch.equal(a, b)


def test_acting_stage_kernel_equals_the_framework_staging():
    """svla_acting_stage (one launch) == what the general path does with ~20 framework ops: static copies of the step's inputs, the T5 padding mask (int64 and uint8, column 0 on),
    kvalid[b, s] = (s <= t) & (s >= max(t - time_step_b, 0)) (allenact_dino_transformer.py:388-397), the device step counter and the int32-wrapping seed bumps."""
    _need_gpu()
    from safevla_amd import ops

    g = torch.Generator(device=DEV).manual_seed(2)
    B, L, MS = 13, 7, 500
    for t in (0, 1, 37, 498):
        tok = torch.randn(B, 2, 84, 384, device=DEV, generator=g).to(BF16)
        pa = torch.randint(0, 20, (1, B), device=DEV, generator=g)
        mk = (torch.rand(1, B, 1, device=DEV, generator=g) > 0.2).float()
        hand = torch.randint(0, 2, (1, B, 1), device=DEV, generator=g)
        ts = torch.randint(0, 600, (1, B), device=DEV, generator=g)
        ids = torch.randint(0, 5, (1, B, L), device=DEV, generator=g)
        d = dict(tok=torch.zeros_like(tok), pa=torch.zeros(B, dtype=torch.int64, device=DEV), mk=torch.zeros(B, device=DEV), hand=torch.zeros(B, dtype=torch.int64, device=DEV),
                 ts=torch.zeros(B, dtype=torch.int64, device=DEV), ids=torch.zeros(B, L, dtype=torch.int64, device=DEV), am=torch.zeros(B, L, dtype=torch.int64, device=DEV),
                 am8=torch.zeros(B, L, dtype=torch.uint8, device=DEV), kv=torch.full((B, MS), 7, dtype=torch.uint8, device=DEV), t_dev=torch.zeros((), dtype=torch.int64, device=DEV))
        seeds = [torch.tensor([v], dtype=torch.int32, device=DEV) for v in (5, 0x7FFFFFF0, -3)]
        want_seeds = [s_.clone().add_(0x3C6EF35) for s_ in seeds]
        ops.acting_stage(tok, d["tok"], pa, d["pa"], mk, d["mk"], hand, d["hand"], ts, d["ts"], ids, d["ids"], d["am"], d["am8"], d["kv"], d["t_dev"], B, L, MS, t, seeds, 0x3C6EF35)
        torch.cuda.synchronize()
        assert torch.equal(d["tok"], tok) and torch.equal(d["pa"], pa.reshape(B)) and torch.equal(d["mk"], mk.reshape(B)) and torch.equal(d["hand"], hand.reshape(B))
        assert torch.equal(d["ts"], ts.reshape(B)) and torch.equal(d["ids"], ids.reshape(B, L)) and int(d["t_dev"]) == t
        am = (ids.reshape(B, L) != 0).to(torch.int64); am[:, 0] = 1
        assert torch.equal(d["am"], am) and torch.equal(d["am8"], am.to(torch.uint8))
        ar = torch.arange(MS, device=DEV)
        kv = ((ar[None, :] <= t) & (ar[None, :] >= torch.clamp(t - ts.reshape(B), min=0)[:, None])).to(torch.uint8)
        assert torch.equal(d["kv"], kv)
        for a, b in zip(seeds, want_seeds):
            assert torch.equal(a, b)


def test_grouped_launches_random_shapes_and_flavours():
    """40 random (M, N, K, epilogue, group size 2 / 3) GEMMs and norms: a capture's result equals the single launches bit for bit whatever kernel the dispatcher picks
    (128-tile, persistent 256-tile, assembly with a 128-tile row tail), and nothing is left in a capture afterwards."""
    _need_gpu()
    from safevla_amd import ops

    rs = np.random.RandomState(2026)
    g = torch.Generator(device=DEV).manual_seed(99)
    for case in range(40):
        members = int(rs.choice([2, 3]))
        M = int(rs.choice([1, 17, 64, 200, 768, 1500, 4096, 11584, 20000]))
        N = int(rs.choice([128, 256, 512, 1024, 1536, 2048]))
        K = int(rs.choice([384, 512, 1024, 2048]))
        flav = rs.choice(["plain", "relu", "res", "res_drop", "relu_drop", "f32"])
        A = [torch.randn(M, K, device=DEV, generator=g).to(BF16) for _ in range(members)]
        W = [(torch.randn(N, K, device=DEV, generator=g) * 0.05).to(BF16) for _ in range(members)]
        b = [torch.randn(N, device=DEV, generator=g) for _ in range(members)]
        R = [torch.randn(M, N, device=DEV, generator=g).to(BF16) for _ in range(members)]

        def call(m):
            kw = {}
            if "relu" in flav: kw["act"] = ops.ACT_RELU
            if "res" in flav: kw["residual"] = R[m]
            if "drop" in flav: kw["drop"] = ops.Dropout(77 + m + case, 2, 0.1)
            if flav == "f32": kw["out_f32"] = True
            return ops.gemm_nt(A[m], W[m], M, N, K, bias=b[m], **kw)

        want = [call(m) for m in range(members)]
        ops.group_stats()
        got = _grouped(call, members)
        torch.cuda.synchronize()
        for m in range(members):
            assert torch.equal(got[m], want[m]), (case, M, N, K, flav, members, m)
        if M >= 8 and N == 512:
            x = [t.clone() for t in got] if flav != "f32" else [t.to(BF16) for t in got]
            gam = [torch.rand(512, device=DEV, generator=g) + 0.5 for _ in range(members)]
            nf = lambda m: ops.norm_fwd(x[m], gam[m], b[m], 1e-5, M, D=512)[0]
            wn = [nf(m) for m in range(members)]
            gn = _grouped(nf, members)
            torch.cuda.synchronize()
            for m in range(members):
                assert torch.equal(gn[m], wn[m]), (case, "norm", M)
    from safevla_amd import _lib
    assert _lib.lib().cdll.svla_group_end(None) != 0          # no capture left open
