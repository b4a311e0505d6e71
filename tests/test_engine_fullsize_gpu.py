"""The ENGINE at BASELINE.json's full sizes (no -m gpu test ran the engine above 18 rows):
C2 = ObjectNav, 32 envs x 128 steps; C3 = PickUp, 64 envs x 256 steps, cost constraint active, 2 env-chunks of 32.
No oracle can run these sizes in seconds, so the checks are size-independent properties: exactness of the env-chunked gradient
accumulation, finiteness, the direction of the lambda update, the per-step bound of Adam, and the loss decreasing on a fixed rollout."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    torch.manual_seed(0)
    return SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)


def test_c2_chunked_accumulation_equals_unchunked_gradient(model):
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.eval()                    # dropout masks are indexed by chunk-local rows: exactness is an eval-mode property
    T, B = 128, 32
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=12, task="ObjectNav", seed=11), device=DEV)
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    grads, sums = {}, {}
    for chunk in (None, 8):
        eng = PPOLagEngine(model, PPOLagConfig(env_chunk=chunk))
        model.zero_grad()
        eng._sums.zero_()
        step = chunk or B
        for c0 in range(0, B, step):
            eng._accumulate(st.batch_slice(c0, c0 + step), T * B, 0.2, last=c0 + step >= B)
        grads[chunk], sums[chunk] = model.arena.flat_g.clone(), eng._sums.clone()
    a, b = grads[None].double(), grads[8].double()
    assert torch.isfinite(a).all() and a.norm().item() > 0
    assert ((a - b).norm() / a.norm()).item() < 5e-3          # bf16 re-rounding of per-chunk partial sums / atomics order only
    assert torch.nn.functional.cosine_similarity(a, b, dim=0).item() > 0.99998
    np.testing.assert_allclose(sums[None].cpu().numpy(), sums[8].cpu().numpy(), rtol=1e-4, atol=1e-6)
    # every tower received a gradient
    for lo, hi in model.arena.tower_ranges:
        assert grads[None][lo:hi].abs().sum().item() > 0


def test_c3_unchunked_pass_equals_chunked_gradient(model):
    """The bench's exact execution: C3 (64 envs x 256 steps = 16 384 rows) in ONE pass per tower (~140 GB of one tower's saved
    activations resident, env_chunk=None) against the same minibatch accumulated over two env-chunks of 32, eval mode."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.eval()
    T, B = 256, 64
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=12, task="PickUp", seed=12), device=DEV)
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    grads, sums = {}, {}
    for chunk in (None, 32):
        eng = PPOLagEngine(model, PPOLagConfig(env_chunk=chunk, cost_limit=2.31964))
        model.zero_grad()
        eng._sums.zero_()
        step = chunk or B
        for c0 in range(0, B, step):
            eng._accumulate(st.batch_slice(c0, c0 + step), T * B, 0.2, last=c0 + step >= B)
        grads[chunk], sums[chunk] = model.arena.flat_g.clone(), eng._sums.clone()
        del eng
        torch.cuda.empty_cache()
    a, b = grads[None].double(), grads[32].double()
    assert torch.isfinite(a).all() and a.norm().item() > 0
    assert torch.nn.functional.cosine_similarity(a, b, dim=0).item() > 0.9999
    assert ((a - b).norm() / a.norm()).item() < 1e-2
    np.testing.assert_allclose(sums[None].cpu().numpy(), sums[32].cpu().numpy(), rtol=1e-4, atol=1e-6)
    for lo, hi in model.arena.tower_ranges:
        assert grads[None][lo:hi].abs().sum().item() > 0
    model.zero_grad()


@pytest.mark.parametrize("T,B,chunk", [(16, 4, None), (128, 32, 16)])
def test_deterministic_mode_gives_bitwise_repeatable_gradients(model, T, B, chunk):
    """PPOLagConfig(deterministic=True): every cross-workgroup gradient accumulation (weight / bias / LayerNorm / embedding / text-feature
    gradients) goes through 64-bit fixed point, so the flat gradient is bitwise identical run to run -- small shapes (three towers on
    concurrent streams, 128-tile kernels) and C2-sized passes (256-tile kernels, two env-chunks) -- and agrees with the fp32-atomic path."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.eval()
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=12, task="PickUp", seed=21), device=DEV)
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])

    def grad(det):
        eng = PPOLagEngine(model, PPOLagConfig(env_chunk=chunk, cost_limit=2.31964, deterministic=det, record_small_updates=False))
        model.zero_grad()
        eng._sums.zero_()
        step = chunk or B
        for c0 in range(0, B, step):
            eng._accumulate(st.batch_slice(c0, c0 + step), T * B, 0.2, last=c0 + step >= B)
        torch.cuda.synchronize()
        g = model.arena.flat_g.clone()
        if det:
            assert int(eng._det_shadow.abs().max().item()) == 0          # folded back and cleared
            from safevla_amd import ops
            assert ops.det_bypass_count(reset=True) == 0                 # no partial left the fixed-point shadow: the run WAS deterministic
        del eng
        return g

    a, b, c = grad(True), grad(True), grad(False)
    assert torch.isfinite(a).all() and a.norm().item() > 0
    assert torch.equal(a, b)
    assert torch.nn.functional.cosine_similarity(a.double(), c.double(), dim=0).item() > 0.999999
    assert ((a.double() - c.double()).norm() / c.double().norm()).item() < 1e-4
    model.zero_grad()


def test_deterministic_mode_repeats_the_whole_update_bitwise(model):
    """Round 5: the squared gradient norm behind the clip coefficient is reduced in a fixed order (svla_sumsq_f32), so with deterministic=True the PARAMETERS
    after a full update (GAE, lambda, 2 epochs of forward / losses / backward / clip / Adam) are bitwise identical from run to run, not only the gradients."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.eval()
    T, B = 64, 16
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=12, task="PickUp", seed=33), device=DEV)
    p0, m0, v0 = model.arena.flat_p.clone(), model.arena.flat_m.clone(), model.arena.flat_v.clone()
    outs = []
    try:
        for _ in range(2):
            model.arena.flat_p.copy_(p0); model.arena.flat_m.copy_(m0); model.arena.flat_v.copy_(v0)
            model.sync_weights(frozen=False)
            eng = PPOLagEngine(model, PPOLagConfig(update_repeats=2, cost_limit=2.31964, deterministic=True, record_small_updates=False))
            info = eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])
            torch.cuda.synchronize()
            assert info["det_bypassed_partials"] == 0
            outs.append(model.arena.flat_p.clone())
            del eng
    finally:
        model.arena.flat_p.copy_(p0); model.arena.flat_m.copy_(m0); model.arena.flat_v.copy_(v0)
        model.sync_weights(frozen=False)
        model.zero_grad()
    assert torch.isfinite(outs[0]).all() and not torch.equal(outs[0], p0)
    assert torch.equal(outs[0], outs[1])


def test_c5_shard_fp8_attention_gradient_close_to_bf16(model):
    """BASELINE configs[4]: one GPU's 32 envs x 256 steps of the mixed-task sampler with 64-token instructions (S = 233), fusion-encoder
    attention on the fp8 MFMA kernels (e4m3 Q/K/V/P, e5m2 dO/dS) against the same minibatch through the bf16 kernels, eval mode.
    Measured (random-init weights): flat-gradient cosine 0.99999, relative L2 distance 0.5 % -- the kernel-level errors of
    tests/test_fp8_attention_gpu.py (4 % on O, 6-9 % on dQ/dK/dV) enter the policy gradient through two of ~40 GEMM-sized terms."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.eval()
    T, B = 256, 32
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=64, task="Mixed", seed=5), device=DEV)
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    grads, sums = {}, {}
    try:
        for fp8 in (False, True):
            model.set_fp8_attention(fp8)
            eng = PPOLagEngine(model, PPOLagConfig(env_chunk=None, cost_limit=2.31964))
            model.zero_grad()
            eng._sums.zero_()
            eng._accumulate(st.batch_slice(0, B), T * B, 0.2, last=True)
            grads[fp8], sums[fp8] = model.arena.flat_g.clone(), eng._sums.clone()
            del eng
            torch.cuda.empty_cache()
    finally:
        model.set_fp8_attention(False)
    a, b = grads[False].double(), grads[True].double()
    assert torch.isfinite(b).all() and b.norm().item() > 0
    cos = torch.nn.functional.cosine_similarity(a, b, dim=0).item()
    rel = ((a - b).norm() / a.norm()).item()
    assert cos > 0.999 and rel < 0.05, (cos, rel)
    np.testing.assert_allclose(sums[True].cpu().numpy(), sums[False].cpu().numpy(), rtol=5e-2, atol=1e-3)   # losses / entropy sums
    for lo, hi in model.arena.tower_ranges:
        assert grads[True][lo:hi].abs().sum().item() > 0
    model.zero_grad()


def test_c3_full_update_lambda_active(model):
    """BASELINE configs[2]: PickUp, 64 envs x 256 steps, cost_limit 2.31964 (README.md:255), env-chunk 32, train mode (dropout on)."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.train()
    T, B = 256, 64
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=12, task="PickUp", seed=12), device=DEV)
    cfg = PPOLagConfig(env_chunk=32, cost_limit=2.31964)
    eng = PPOLagEngine(model, cfg)
    p0 = model.arena.flat_p.clone()
    jc = ep["episode_cost_sum"] / ep["n_episodes"]
    assert jc > cfg.cost_limit                                  # Binomial(5, 0.05) per step x ~42-step episodes: the constraint is violated
    info = eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])
    assert info["env_steps"] == T * B == 16384 and eng.opt_step == cfg.update_repeats and eng.tower_steps == [4, 4, 4]
    assert all(np.isfinite(v) for v in info.values())
    assert abs(info["Jc"] - jc) < 1e-9 and info["lagrangian_multiplier"] > cfg.lambda_init      # lambda moves toward the constraint
    d = (model.arena.flat_p - p0).abs().max().item()
    assert 0 < d <= cfg.update_repeats * cfg.lr * 1.01           # |Adam step| <= lr per optimiser step
    assert torch.isfinite(model.arena.flat_p).all()
    # bf16 mirrors in sync with the fp32 masters after the update
    assert torch.equal(model.arena.flat_bf16, model.arena.flat_p.to(torch.bfloat16))
    # more updates on the same rollout (eval mode: no dropout noise in the reported losses): lambda keeps rising (Jc fixed above the
    # limit) and the critics' losses fall
    model.eval()
    infos = [eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"]) for _ in range(3)]
    assert infos[2]["lagrangian_multiplier"] > infos[0]["lagrangian_multiplier"] > info["lagrangian_multiplier"]
    assert infos[2]["value"] < infos[0]["value"] and infos[2]["c_value"] < infos[0]["c_value"]
    model.arena.flat_p.copy_(p0)
    model.sync_weights(frozen=False)


def test_c5_shard_mixed_tasks_long_instructions(model):
    """One GPU's shard of BASELINE configs[4]: mixed ObjectNav / PickUp / Fetch envs (env e -> task e mod 3), 64-token instructions
    (S = 233 fusion tokens), reduced to 8 envs x 64 steps: one full update is finite and moves every tower."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.train()
    T, B = 64, 8
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=64, task="Mixed", seed=13, env_offset=32), device=DEV)
    hand = st.observations["an_object_is_in_hand"][:, :, 0].float().mean(0)       # ObjectNav envs never hold an object
    assert hand[[1, 4, 7]].sum().item() == 0                                      # (32 + b) % 3 == 0 -> ObjectNav
    eng = PPOLagEngine(model, PPOLagConfig(update_repeats=1, env_chunk=4))
    p0 = model.arena.flat_p.clone()
    info = eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])
    assert all(np.isfinite(v) for v in info.values())
    for lo, hi in model.arena.tower_ranges:
        assert not torch.equal(model.arena.flat_p[lo:hi], p0[lo:hi])
    model.arena.flat_p.copy_(p0)
    model.sync_weights(frozen=False)


def test_c4_shard_fetch_chunked_equals_unchunked_and_lambda_moves(model):
    """BASELINE configs[3]'s single-GPU half: ONE rank's shard of the 256-env Fetch run over 8 GPUs = 32 envs x 256 steps (env_offset: envs 96..127
    of the global run).  (a) eval mode: the one-pass gradient (the bench's execution: the assembly GEMMs of asmgen/ carry it at this size) equals the
    accumulation over two env-chunks of 16; (b) train mode: one full update with the cost constraint active is finite, moves every tower, and lambda
    rises (Jc above the limit).  The 8-GPU half (gradient / cost all-reduce) is covered at world sizes 2 and 8 by tests/test_dp_gpu.py and the bench tests."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.eval()
    T, B = 256, 32
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=12, task="Fetch", seed=14, env_offset=96), device=DEV)
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    grads, sums = {}, {}
    for chunk in (None, 16):
        eng = PPOLagEngine(model, PPOLagConfig(env_chunk=chunk, cost_limit=2.31964))
        model.zero_grad()
        eng._sums.zero_()
        step = chunk or B
        for c0 in range(0, B, step):
            eng._accumulate(st.batch_slice(c0, c0 + step), T * B, 0.2, last=c0 + step >= B)
        grads[chunk], sums[chunk] = model.arena.flat_g.clone(), eng._sums.clone()
        del eng
        torch.cuda.empty_cache()
    a, b = grads[None].double(), grads[16].double()
    assert torch.isfinite(a).all() and a.norm().item() > 0
    assert torch.nn.functional.cosine_similarity(a, b, dim=0).item() > 0.9999
    assert ((a - b).norm() / a.norm()).item() < 1e-2
    np.testing.assert_allclose(sums[None].cpu().numpy(), sums[16].cpu().numpy(), rtol=1e-4, atol=1e-6)
    model.zero_grad()
    model.train()
    cfg = PPOLagConfig(env_chunk=None, cost_limit=2.31964)
    eng = PPOLagEngine(model, cfg)
    p0 = model.arena.flat_p.clone()
    assert ep["episode_cost_sum"] / ep["n_episodes"] > cfg.cost_limit
    info = eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])
    assert info["env_steps"] == T * B and all(np.isfinite(v) for v in info.values())
    assert info["lagrangian_multiplier"] > cfg.lambda_init
    for lo, hi in model.arena.tower_ranges:
        assert not torch.equal(model.arena.flat_p[lo:hi], p0[lo:hi])
    model.arena.flat_p.copy_(p0)
    model.sync_weights(frozen=False)


def test_det_bypass_counter_counts_partials_that_leave_the_shadow():
    """a partial sum with |partial| >= 0.25 (or NaN / Inf) skips the fixed-point shadow and takes the plain fp32 atomic -- the run is then not bitwise
    repeatable.  svla_det_bypass_count makes that visible: 0 for ordinary magnitudes, > 0 (and the fp32 result still right) when the partials are large."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from safevla_amd import ops

    M, N = 4096, 512
    for scale, want_bypass in ((1e-4, False), (50.0, True)):
        dY = (torch.ones(M, N, device=DEV) * scale).to(torch.bfloat16)
        db = torch.zeros(N, device=DEV)
        shadow = torch.zeros(N, device=DEV, dtype=torch.int64)
        ops.det_bypass_count(reset=True)
        ops.det_config(0, db, shadow)
        try:
            ops.colsum_acc(dY, db, M, N)
            ops.det_finalize(db, shadow)
        finally:
            ops.det_config(0, None, None)
        n = ops.det_bypass_count(reset=True)
        assert (n > 0) == want_bypass, (scale, n)
        want = dY.float().sum(0)
        assert torch.allclose(db, want, rtol=1e-5, atol=0), (scale, float((db - want).abs().max()))
