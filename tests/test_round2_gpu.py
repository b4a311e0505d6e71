"""Round-2 additions on the GPU: HL-Gauss discrete critic (kernel vs the reference-generated golden G1, head + loss through the
model API and the engine), clipped value loss, fp32 strided GEMM, the model's ``extras``, per-tower Adam bookkeeping, normalised
advantages, the steppable synthetic vector env / acting-path rollout collection, Lightning checkpoints through ``build_agent``."""
import math
import os
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from safevla_amd import ops as o

    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------------ HL-Gauss kernel
def test_hlgauss_kernel_vs_reference_golden(ops):
    """G1 was produced by the reference's own HLGaussLoss (tests/golden/make_golden.py): probs, loss and read-out value."""
    g = dict(np.load(os.path.join(G, "g1_hlgauss.npz")))
    logits, target = torch.from_numpy(g["logits"]).to(DEV), torch.from_numpy(g["target"]).to(DEV)
    R, NB = logits.shape
    values, dl, sums = ops.hlgauss_fwd_bwd(logits, target, None, -5.0, 15.0, 0.15, 1.0, 1.0 / R)
    np.testing.assert_allclose(sums.item() / R, float(g["loss"]), rtol=2e-5)
    # the reference's read-out: transform_from_probs(softmax(logits)) (golden "value")
    np.testing.assert_allclose(values.cpu().numpy(), g["value"], rtol=1e-5, atol=1e-5)
    # gradient of the mean cross-entropy against the reference's probabilities: (softmax - q) / R
    want_dl = (torch.softmax(torch.from_numpy(g["logits"]), -1) - torch.from_numpy(g["probs"])) / R
    np.testing.assert_allclose(dl.cpu().numpy(), want_dl.numpy(), rtol=1e-4, atol=2e-7)
    # transform_from_probs(transform_to_probs(t)) ~ t (bin width 0.2, sigma 0.15): feed log(probs) as logits -> softmax == probs
    lp = torch.log(torch.from_numpy(g["probs"]).clamp_min(1e-30)).to(DEV)
    v2, _, _ = ops.hlgauss_fwd_bwd(lp, None, None, -5.0, 15.0, 0.15, want_grad=False)
    np.testing.assert_allclose(v2.cpu().numpy(), g["target"], atol=2e-2)


def test_hlgauss_value_path_gradient_and_loss_object(ops):
    from oracle import ref_loss
    from safevla_amd.losses import HLGaussLoss

    R, NB = 37, 101
    logits, target, dval = rnd(R, NB, seed=1), rnd(R, seed=2, scale=3.0) + 4.0, rnd(R, seed=3)
    sup = ref_loss.hl_support()
    x = logits.clone().requires_grad_(True)
    loss = 0.5 * 0.7 * ref_loss.hl_gauss_loss(x, target, sup) + (ref_loss.hl_gauss_value(torch.softmax(x, -1), sup) * dval).sum()
    loss.backward()
    _, dl, sums = ops.hlgauss_fwd_bwd(logits.to(DEV), target.to(DEV), dval.to(DEV), -5.0, 15.0, 0.15, 0.5 * 0.7, 1.0 / R)
    np.testing.assert_allclose(dl.cpu().numpy(), x.grad.numpy(), rtol=2e-4, atol=1e-6)
    # the loss object (reference API: loss_fn(logits, target), transform_from_probs) on device tensors == host closed form
    hl = HLGaussLoss(-5.0, 15.0, 101, 0.15)
    xg = logits.to(DEV).requires_grad_(True)
    l = hl(xg, target.to(DEV))
    l.backward()
    want = ref_loss.hl_gauss_loss(logits, target, sup)
    np.testing.assert_allclose(l.item(), want.item(), rtol=2e-5)
    x2 = logits.clone().requires_grad_(True)
    ref_loss.hl_gauss_loss(x2, target, sup).backward()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), x2.grad.numpy(), rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(hl.value_from_logits(logits.to(DEV)).cpu().numpy(),
                               ref_loss.hl_gauss_value(torch.softmax(logits, -1), sup).numpy(), rtol=1e-5, atol=1e-5)


def test_value_mse_clipped_matches_reference_expression(ops):
    """customized_loss.py:374-380 (the same expression upstream PPOValue uses)."""
    R, clip = 1000, 0.1
    v, r, ov = rnd(R, seed=1), rnd(R, seed=2), rnd(R, seed=3)
    ov = v + 0.3 * ov                       # a mix of clipped and unclipped rows
    vv = v.clone().requires_grad_(True)
    vc = ov + (vv - ov).clamp(-clip, clip)
    l = 0.5 * torch.max((vv - r).pow(2), (vc - r).pow(2)).mean()
    l.backward()
    sums, dv = ops.value_mse_fwd_bwd(v.to(DEV), r.to(DEV), 1.0, 1.0 / R, old_values=ov.to(DEV), clip=clip)
    np.testing.assert_allclose(0.5 * sums.item() / R, l.item(), rtol=1e-5)
    np.testing.assert_allclose(dv.cpu().numpy(), vv.grad.numpy(), rtol=1e-5, atol=1e-9)
    from safevla_amd.api import SafeActorCriticOutput
    from safevla_amd.losses import PPOValue

    vg = v.to(DEV).view(R, 1, 1).requires_grad_(True)
    aco = SafeActorCriticOutput(distributions=None, values=vg, c_values=None)
    tot, info = PPOValue(clip_param=clip, use_clipped_value_loss=True).loss(0, {"values": ov.to(DEV).view(R, 1, 1), "returns": r.to(DEV).view(R, 1, 1)}, aco)
    tot.backward()
    np.testing.assert_allclose(info["value"], l.item(), rtol=1e-5)
    np.testing.assert_allclose(vg.grad.reshape(R).cpu().numpy(), vv.grad.numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("M,N,K", [(70, 101, 256), (513, 256, 512), (5, 1, 33)])
def test_gemm_f32_all_layouts(ops, M, N, K):
    A, B, bias, res = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    want = F.relu(A @ B.t() + bias) + res
    got = ops.gemm_f32(A.to(DEV), B.to(DEV), M, N, K, bias=bias.to(DEV), act=ops.ACT_RELU, residual=res.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-4)      # fp32 accumulation order (K up to 512)
    # NN: dX = dY . W  (B given as [K, N] row-major)
    Bt = B.t().contiguous()
    got = ops.gemm_f32(A.to(DEV), Bt.to(DEV), M, N, K, sb=(1, N))
    np.testing.assert_allclose(got.cpu().numpy(), (A @ B.t()).numpy(), rtol=1e-4, atol=1e-4)
    # TN with accumulate: dW[N, K] += dY[M, N]^T X[M, K]
    dY, X = rnd(M, N, seed=5), rnd(M, K, seed=6)
    acc = rnd(N, K, seed=7)
    out = acc.clone().to(DEV)
    ops.gemm_f32(dY.to(DEV), X.to(DEV), N, K, M, sa=(1, N), sb=(1, K), out=out, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), (acc + dY.t() @ X).numpy(), rtol=1e-4, atol=1e-4)
    # ReLU mask + colsum
    mask = rnd(M, N, seed=8)
    got = ops.gemm_f32(A.to(DEV), B.to(DEV), M, N, K, mask=mask.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), ((A @ B.t()) * (mask > 0)).numpy(), rtol=1e-4, atol=1e-4)
    cs = torch.zeros(N, device=DEV)
    ops.colsum_f32(dY.to(DEV), cs, M, N)
    np.testing.assert_allclose(cs.cpu().numpy(), dY.sum(0).numpy(), rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------------------------------------ model: discrete critic, extras
def _obs(T, B, seed=0, L=6):
    rs = np.random.RandomState(seed)
    ids = rs.randint(3, 32000, size=(B, L)); ids[:, -1] = 1
    return {
        "rgb_dinov2": torch.from_numpy(rs.standard_normal((T, B, 384, 7, 12)).astype(np.float32)).to(DEV),
        "manipulation_rgb_dinov2": torch.from_numpy(rs.standard_normal((T, B, 384, 7, 12)).astype(np.float32)).to(DEV),
        "goal_token_ids": torch.from_numpy(np.broadcast_to(ids, (T, B, L)).copy()).to(DEV),
        "time_step": torch.arange(T)[:, None].expand(T, B).contiguous().to(DEV),
        "traj_index": torch.zeros(T, B, dtype=torch.int64, device=DEV),
        "an_object_is_in_hand": torch.zeros(T, B, 1, dtype=torch.int64, device=DEV),
    }, torch.from_numpy(rs.randint(0, 20, size=(T, B))).to(DEV), torch.ones(T, B, 1, device=DEV)


@pytest.mark.parametrize("critic_type", ["discrete", "mlp"])
def test_critic_heads_forward_backward_vs_torch(ops, critic_type):
    """DiscreteCriticHead / MLPCriticHead (allenact_dino_transformer.py:720-766): the head's forward and every parameter / input
    gradient against a torch.autograd restatement on the same beliefs."""
    from oracle import ref_loss
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    torch.manual_seed(0)
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, critic_type=critic_type).eval()
    tw = m.critic_tsfm
    names = [k for k in m.state_dict() if k.startswith("critic_tsfm.critic.fc.")]
    assert names == ([f"critic_tsfm.critic.fc.{i}.{p}" for i in (0, 2) for p in ("weight", "bias")] if critic_type == "discrete"
                     else [f"critic_tsfm.critic.fc.{i}.{p}" for i in (0, 2, 4) for p in ("weight", "bias")])
    T, B = 5, 3
    obs, pa, mk = _obs(T, B)
    m.zero_grad()
    prep = m.prepare(obs, pa, mk)
    _, values, c = tw.run_forward(prep, need_grad=True)
    beliefs = c["beliefs"].clone()                                   # [R, 512] fp32, rows b*T + t
    R = T * B
    dval = rnd(T, B, 1, seed=5).to(DEV)
    dfull = rnd(T, B, 101, seed=6, scale=0.1).to(DEV) if critic_type == "discrete" else None
    full = tw._last_full_logits
    tw.run_backward(prep, c, None, dval, dfull)
    # torch restatement
    fc = tw.critic.fc
    ws = [(fc[i].weight.detach().clone().requires_grad_(True), fc[i].bias.detach().clone().requires_grad_(True)) for i in range(0, len(fc), 2)]
    x = beliefs.clone().requires_grad_(True)
    h = x
    for j, (w, b) in enumerate(ws):
        h = F.linear(h, w, b)
        if j < len(ws) - 1:
            h = F.relu(h)
    out_tb = h.view(B, T, -1).transpose(0, 1)
    if critic_type == "discrete":
        sup = ref_loss.hl_support().to(DEV)
        v_ref = ref_loss.hl_gauss_value(torch.softmax(out_tb, -1), sup).unsqueeze(-1)
        np.testing.assert_allclose(full.cpu().numpy(), out_tb.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
        ((v_ref * dval).sum() + (out_tb * dfull).sum()).backward()
    else:
        v_ref = out_tb
        (v_ref * dval).sum().backward()
    np.testing.assert_allclose(values.cpu().numpy(), v_ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    for i, (w, b) in zip(range(0, len(fc), 2), ws):
        np.testing.assert_allclose(tw.g(fc[i].weight).cpu().numpy(), w.grad.cpu().numpy(), rtol=2e-4, atol=2e-6)
        np.testing.assert_allclose(tw.g(fc[i].bias).cpu().numpy(), b.grad.cpu().numpy(), rtol=2e-4, atol=2e-6)
    # the gradient reached the decoder: dW of the decoder output projection is non-zero and finite
    gw = tw.g(tw.decoder.output.weight)
    assert torch.isfinite(gw).all() and gw.abs().sum().item() > 0


def test_discrete_critics_through_the_reference_api_and_engine(ops):
    """SafePPOLogGrad(discrete_critics=True): value term = 0.5 * loss_func(extras["full_logits"], returns)
    (customized_loss.py:364-370); the extras are the cost-critic tower's (separate_actor_critic.py:31-36).  API path
    (``total.backward()``) and engine path must produce the same gradients, and both match the oracle formulas on the model's outputs."""
    from oracle import ref_loss
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.losses import SafePPOLogGrad
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    torch.manual_seed(0)
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, critic_type="discrete").eval()
    T, B = 4, 3
    obs, pa, mk = _obs(T, B, seed=2)
    rs = np.random.RandomState(3)
    batch = {"actions": torch.from_numpy(rs.randint(0, 20, size=(T, B))).to(DEV), "old_action_log_probs": torch.full((T, B), -3.0, device=DEV),
             "adv_targ": rnd(T, B, 1, seed=1).to(DEV), "c_adv_targ": rnd(T, B, 1, seed=2).to(DEV), "returns": (rnd(T, B, 1, seed=3) * 2 + 3).to(DEV),
             "values": rnd(T, B, 1, seed=4).to(DEV), "c_returns": rnd(T, B, 1, seed=5).to(DEV)}
    m.zero_grad()
    aco, _ = m(obs, None, pa, mk)
    assert {"full_logits", "loss_func", "stop_grad_logits", "total_norm", "weight_norm", "bias_norm", "weight_grad_norm"} <= set(aco.extras)
    assert aco.extras["full_logits"].shape == (T, B, 101)
    loss = SafePPOLogGrad(0.1, 0.5, 0.0, use_clipped_value_loss=False, discrete_critics=True, normalize_advantage=False)
    total, info = loss.loss(0, batch, aco, lagrangian_multiplier=torch.tensor(0.37))
    # oracle formulas on the model's own outputs
    cb = {k: v.cpu() for k, v in batch.items()}
    _, ri = ref_loss.safe_ppo_log_grad(aco.distributions.raw_logits.detach().cpu(), aco.values.detach().cpu(), cb, 0.37)
    want_v = 0.5 * ref_loss.hl_gauss_loss(aco.extras["full_logits"].detach().cpu().view(-1, 101), cb["returns"].view(-1), ref_loss.hl_support())
    np.testing.assert_allclose(info["value"], want_v.item(), rtol=1e-4)
    np.testing.assert_allclose(info["action"], ri["action"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(info["ppo_total"], 0.5 * want_v.item() + ri["action"], rtol=1e-4, atol=1e-6)
    total.backward()
    g_api = m.arena.flat_g.clone()
    a1, b1 = m.arena.tower_ranges[1]
    assert g_api[a1:b1].abs().sum().item() == 0        # the reward-critic tower gets no gradient from this loss (reference data flow)
    eng = PPOLagEngine(m, PPOLagConfig(stage_losses=("ppo_log_loss",)))
    assert eng.active_towers() == (True, False, True)
    m.zero_grad()
    eng._sums.zero_()
    eng._accumulate({**batch, "observations": obs, "prev_actions": pa, "masks": mk}, T * B, 0.37)
    g_eng = m.arena.flat_g
    err = (g_eng - g_api).norm() / g_api.norm()
    assert err.item() < 2e-3, err.item()               # same kernels, atomics order only
    np.testing.assert_allclose(0.5 * eng._sums[3].item() / (T * B), want_v.item(), rtol=1e-4)


def test_extras_keys_of_the_default_model(ops):
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV).eval()
    obs, pa, mk = _obs(3, 2)
    aco, _ = m(obs, None, pa, mk)
    ex = aco.extras
    assert {"total_norm", "stop_grad_values", "weight_norm", "bias_norm", "weight_grad_norm"} <= set(ex)
    assert torch.equal(ex["stop_grad_values"], aco.c_values.detach()) and not ex["stop_grad_values"].requires_grad
    fc = m.c_critic_tsfm.critic.fc
    np.testing.assert_allclose(ex["weight_norm"].item(), fc.weight.norm(2).item(), rtol=1e-6)
    assert ex["total_norm"].item() == 0.0 and ex["weight_grad_norm"].item() == 0.0       # no backward yet
    aco.c_values.sum().backward()
    aco2, _ = m(obs, None, pa, mk)
    a, b = m.arena.tower_ranges[2]
    np.testing.assert_allclose(aco2.extras["total_norm"].item(), m.arena.flat_g[a:b].double().norm().item(), rtol=1e-5)
    assert aco2.extras["weight_grad_norm"].item() > 0
    # arena layout: every tower starts on a 256-element boundary (16-byte aligned bf16 weight views)
    assert all(s % 256 == 0 for s, _ in m.arena.tower_ranges) and m.arena.tower_ranges[0][0] == 0
    for t in m.towers:
        for w in t._w.values():
            assert w.data_ptr() % 16 == 0


# ------------------------------------------------------------------------------------------------ engine bookkeeping
def test_adam_skips_towers_without_a_loss_like_torch_optim(ops):
    """torch.optim.Adam after zero_grad(set_to_none=True): parameters of a tower with no loss in the stage keep grad None -> skipped,
    moments frozen, per-parameter step not advanced.  Stage 0 trains the two critics, stage 1+ the policy and the critic."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    torch.manual_seed(0)
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV).eval()
    st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=6, B=2, L=5, task="PickUp", seed=3), device=DEV)
    cfg = PPOLagConfig(update_repeats=1, stage_losses=("ppo_value_loss", "safe_ppo_value_loss"))
    eng = PPOLagEngine(m, cfg)
    ar = m.arena
    p0 = ar.flat_p.clone()
    eng.update(st, nxt["next_value"], nxt["next_c_value"], 3.0, 1.0)
    (a0, b0), (a1, b1), (a2, b2) = ar.tower_ranges
    assert eng.tower_steps == [0, 1, 1] and eng.active_towers() == (False, True, True)
    assert torch.equal(ar.flat_p[a0:b0], p0[a0:b0]) and ar.flat_m[a0:b0].abs().sum().item() == 0
    assert not torch.equal(ar.flat_p[a1:b1], p0[a1:b1]) and not torch.equal(ar.flat_p[a2:b2], p0[a2:b2])
    cfg.stage_losses = ("ppo_log_loss",)                     # the shipped stage 1+: cost critic has no loss any more
    m2, v2, p2 = ar.flat_m[a2:b2].clone(), ar.flat_v[a2:b2].clone(), ar.flat_p[a2:b2].clone()
    eng.update(st, nxt["next_value"], nxt["next_c_value"], 3.0, 1.0)
    assert eng.tower_steps == [1, 2, 1]
    assert torch.equal(ar.flat_p[a2:b2], p2) and torch.equal(ar.flat_m[a2:b2], m2) and torch.equal(ar.flat_v[a2:b2], v2)
    # first Adam step of the policy tower uses ITS step count (1): |delta| == lr wherever the gradient is non-zero
    d = (ar.flat_p[a0:b0] - p0[a0:b0]).abs()
    assert d.max().item() <= 2e-5 * 1.001 and d.max().item() > 2e-5 * 0.99


def test_normalized_advantages(ops):
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV).eval()
    st, nxt, _ = fill_synthetic_rollout(m, SynthSpec(T=9, B=3, L=5, seed=1), device=DEV)
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    b = st.batch_slice(0, 3)
    assert "norm_adv_targ" not in b                       # raw advantages are never passed off as normalised ones
    bn = st.batch_slice(1, 3, normalized=True)
    adv = st.adv_targ
    want = (adv - adv.mean()) / (adv.std() + 1e-5)
    np.testing.assert_allclose(bn["norm_adv_targ"].cpu().numpy(), want[:, 1:3].cpu().numpy(), rtol=1e-4, atol=1e-6)
    cadv = st.c_adv_targ
    np.testing.assert_allclose(bn["c_norm_adv_targ"].cpu().numpy(), ((cadv - cadv.mean()) / (cadv.std() + 1e-5))[:, 1:3].cpu().numpy(), rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ rollout collection through the acting path
def test_vector_env_and_acting_rollout_collection(ops):
    """SynthVectorEnv.step -> SafeRLStepResult contract; collect_rollout fills the storage through the KV-cached single-step policy and
    the stored log-probs / values equal a full-sequence (update-path) forward over the collected rollout (eval mode, equal-length goals)."""
    from safevla_amd.api import SafeRLStepResult
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.storage import RolloutStorage
    from safevla_amd.synth_env import SynthVectorEnv, collect_rollout, env_tasks

    assert env_tasks("Mixed", 5, env_offset=2) == ["Fetch", "ObjectNav", "PickUp", "Fetch", "ObjectNav"]
    torch.manual_seed(0)
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV).eval()
    B, T = 3, 7
    env = SynthVectorEnv(B, L=6, task="Mixed", seed=5, device=DEV)
    obs, r, c, d, res = env.step(torch.tensor([0, 4, 1], device=DEV), want_results=True)
    assert isinstance(res[0], SafeRLStepResult) and res[1].done is True and res[1].info["task_type"] == "PickUp"
    assert 0 <= res[0].cost <= 5 and res[0].reward in (0.0, 10.0) and set(res[0].observation) == set(obs)
    assert int(obs["time_step"][1]) == 0                       # env 1 ended: fresh episode
    s_, n_ = env.pop_episode_costs()
    assert n_ >= 1
    st = RolloutStorage(T, device=DEV)
    nxt = collect_rollout(m, env, st, T, obs0=env.reset())
    assert st.step == T and nxt["next_value"].shape == (B, 1)
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    batch = st.batch_slice(0, B)
    with torch.no_grad():
        out, _ = m(batch["observations"], None, batch["prev_actions"], batch["masks"])
    lp = out.distributions.log_prob(batch["actions"])
    # acting path (KV cache, episode window) == update path (block-causal over traj ids): bf16 kernels on different schedules
    assert (lp - st.action_log_probs).abs().max().item() < 3e-2
    assert (out.values[:T] - st.value_preds[:T]).abs().max().item() < 3e-2 * max(1.0, st.value_preds.abs().max().item())
    st.after_updates()
    assert st.step == 0


def test_build_agent_loads_a_lightning_checkpoint(ops, tmp_path):
    """the Lightning (IL) branch of build_agent used to index ["state_dict"] twice."""
    from safevla_amd.agent import InferenceAgentVIDA
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)
    il = {"state_dict": {"model.actor.weight": torch.full((20, 512), 0.125), "model.actor.bias": torch.full((20,), -2.0),
                         "model.decoder.norm.weight": torch.full((512,), 1.5),
                         "model.visual_encoder.image_encoder.model.norm.weight": torch.full((384,), 2.0)}}
    path = str(tmp_path / "il.ckpt")
    torch.save(il, path)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        agent = InferenceAgentVIDA.build_agent(m, device=DEV, ckpt_path=path)
    assert (m.actor.linear.weight == 0.125).all() and (m.actor.linear.bias == -2).all() and (m.decoder.norm.weight == 1.5).all()
    assert (agent.nav_pre.vit.norm.weight == 2.0).all()                               # image-encoder keys fill the frozen ViT
    msgs = " ".join(str(x.message) for x in w)
    assert "word-hash goal tokenizer" in msgs and "random-init DINOv2" not in msgs    # ViT came from the checkpoint, tokenizer is a stand-in
    assert not m.training


def test_reference_path_entry_point_fire_style(ops, tmp_path):
    """`python training/online/dinov2_vits_tsfm_base.py train --flag value --flag=value` (scripts/train.sh:116-136): same path, same flags;
    collects rollouts through the acting path on the synthetic PickUp env, runs stage-0 updates, writes an AllenAct-style checkpoint."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "training", "online", "dinov2_vits_tsfm_base.py"), "--num_train_processes=2", "train",
           "--output_dir", str(tmp_path), "--dataset_dir", "data/fifteen/PickupType", "--cost_limit", "2.31964", "--tag=PickupType",
           "--num_steps", "8", "--total_steps", "32", "--save_interval", "32", "--seed", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and lines[-1]["training_step"] == 32 and lines[0]["stage"] == ["ppo_value_loss", "safe_ppo_value_loss"]
    assert all(np.isfinite(l["value"]) and np.isfinite(l["c_value"]) for l in lines)
    ck = [f for f in os.listdir(tmp_path) if f.endswith(".pt")]
    assert len(ck) == 1 and "PickupType" in ck[0]
    sd = torch.load(os.path.join(tmp_path, ck[0]), map_location="cpu")
    assert "model_state_dict" in sd and sd["optimizer_state"]["tower_steps"] == [0, 8, 8]      # 2 updates x 4 epochs, critics only


def test_sentencepiece_goal_path_on_the_gpu(ops, tmp_path):
    """byte-string goals -> sentencepiece ids (a model trained here; t5-small's own vocabulary is a network asset)
    -> frozen T5 -> text adapter, de-duplicated per unique string: rows with the same instruction get identical outputs, different
    instructions differ, and the ids that reach the encoder are the sentencepiece ids + EOS."""
    import sentencepiece as spm

    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.text import GoalTokenizer, str_to_bytes

    corpus = tmp_path / "c.txt"
    words = "find a mug pick up the bowl fetch red apple go to sofa navigate locate plate cup laptop".split()
    corpus.write_text("\n".join(" ".join(np.random.RandomState(i).choice(words, 5)) for i in range(400)))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "sp"), vocab_size=40, model_type="unigram",
                                   pad_id=0, eos_id=1, unk_id=2, bos_id=-1)
    tok = GoalTokenizer(str(tmp_path / "sp.model"))
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, tokenizer=tok).eval()
    from oracle.detfill import fill_state_dict
    fill_state_dict(m, seed=5)          # O(1) outputs (the default initialisation gives near-zero logits: nothing to compare relative errors on)
    m.sync_weights()
    T, B = 3, 4
    goals = ["find a mug", "pick up the bowl", "find a mug", "fetch red apple"]
    obs, pa, mk = _obs(T, B)
    del obs["goal_token_ids"]
    obs["natural_language_spec"] = torch.from_numpy(np.stack([np.stack([str_to_bytes(g).reshape(-1) for g in goals])] * T)).to(DEV)
    same = obs["rgb_dinov2"][:, 0].clone()
    obs["rgb_dinov2"][:, 2] = same
    obs["manipulation_rgb_dinov2"][:, 2] = obs["manipulation_rgb_dinov2"][:, 0]
    pa[:, 2] = pa[:, 0]
    prep = m.prepare(obs, pa, mk)
    assert prep.U == 3 and prep.L == max(len(tok.encode(g)) for g in goals)
    want = {tuple(tok.encode(g)) for g in goals}
    got = {tuple(int(x) for x in row if x != 0) for row in prep.ids.cpu().tolist()}
    assert got == want and all(row[-1] == 1 for row in want)
    with torch.no_grad():
        aco, _ = m(obs, None, pa, mk)
    lg = aco.distributions.logits
    assert torch.equal(lg[:, 0], lg[:, 2]) and not torch.equal(lg[:, 0], lg[:, 1])
    # the same byte strings through the CPU oracle with the SAME sentencepiece tokenizer and weights (bytes -> ids -> frozen T5 -> text adapter
    # -> three towers): the sentencepiece branch is now compared with the restatement, not only checked for identity / separation of rows
    from oracle import ref_model

    ref = ref_model.RefSafeActorCritic(tok, max_batch=B).eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        want, _ = ref({k: v.cpu() for k, v in obs.items()}, None, pa.cpu(), mk.cpu())
    # (aco.distributions.logits are the normalised log-probabilities of CategoricalDistr; the oracle returns the raw head outputs)
    for name, got_t, want_t in (("log-probs", lg, torch.log_softmax(want["logits"].float(), -1)), ("values", aco.values, want["values"]), ("c_values", aco.c_values, want["c_values"])):
        err = float((got_t.float().cpu() - want_t.float()).abs().max() / (want_t.float().abs().max() + 1e-12))
        assert err < 4e-2, (name, err)          # bf16 activations vs the fp32 oracle on a 12-row batch (the ladder of DESIGN section 5; measured 2.2e-2)


@pytest.mark.parametrize("N,K,epi", [(384, 384, "res"), (1152, 384, "plain"), (384, 1536, "gelu_res")])
def test_gemm_nt_256_tile_with_half_last_n_tile(ops, N, K, epi):
    """ViT-S widths (384, 1152 = N % 256 == 128): the persistent 256-tile kernel with a half last n-tile == the 128-tile kernel == torch."""
    M = 128 * 433 + 77
    A = rnd(M, K, seed=1).to(torch.bfloat16)
    W = (rnd(N, K, seed=2) / math.sqrt(K)).to(torch.bfloat16)
    bias = rnd(N, seed=3)
    res = rnd(M, N, seed=4).to(torch.bfloat16)
    kw = dict(bias=bias.to(DEV))
    if "res" in epi:
        kw["residual"] = res.to(DEV)
    if "gelu" in epi:
        kw["act"] = ops.ACT_GELU
    big = ops.gemm_nt(A.to(DEV), W.to(DEV), M, N, K, **kw)
    ops.gemm_force_small_tile(True)
    try:
        small = ops.gemm_nt(A.to(DEV), W.to(DEV), M, N, K, **kw)
    finally:
        ops.gemm_force_small_tile(False)
    y = A.float() @ W.float().t() + bias
    if "gelu" in epi:
        y = F.gelu(y)
    if "res" in epi:
        y = y + res.float()
    err = (big.float().cpu() - y).abs().max().item() / y.abs().max().item()
    assert err < 1e-2, err
    assert (big.float() - small.float()).abs().max().item() <= 2e-2 * y.abs().max().item()


def test_recorded_acting_step_equals_eager_acting(ops):
    """The default acting path: the single-step 3-tower forward recorded once as three ops.LaunchPlans (one per tower / HIP stream; step
    counter, KV slot and dropout seed in device memory, attention over the whole cache window behind the mask) must reproduce the eager
    acting path step for step across an episode boundary, survive sampler_select (new caches => new plans) and draw fresh noise in train mode."""
    from oracle.detfill import fill_state_dict
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    g = dict(np.load(os.path.join(G, "g5_samelen.npz"), allow_pickle=False))
    obs = {k[4:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("obs:")}
    pa, mk = torch.from_numpy(g["prev_actions"]).to(DEV), torch.from_numpy(g["masks"]).to(DEV)
    T = pa.shape[0]
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)
    assert m._acting_graphs is not None and m._acting_backend == "plan"          # on by default
    fill_state_dict(m, seed=7)
    m.sync_weights()
    m.eval()

    def run(plans):
        for t in m.towers:
            t.time_step_counter, t._kv = 0, None
        m.enable_acting_plans(plans)
        out = []
        with torch.no_grad():
            for t in range(T):
                o, _ = m({k: v[t:t + 1] for k, v in obs.items()}, None, pa[t:t + 1], mk[t:t + 1])
                out.append((o.distributions.logits.float().cpu(), o.values.cpu(), o.c_values.cpu()))
        return out

    eager, plan = run(False), run(True)
    assert len(next(iter(m._acting_graphs.values())).plans) == 3
    for t, (a, b) in enumerate(zip(eager, plan)):
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=0, atol=2e-2 * max(1.0, x.abs().max().item())), (t, (x - y).abs().max())
    assert all(t.time_step_counter == T for t in m.towers)
    # and against the reference's own step-by-step outputs
    ga = dict(np.load(os.path.join(G, "g5_acting.npz"), allow_pickle=False))
    relm = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    assert relm(torch.cat([p[0] for p in plan]).numpy(), ga["logits"]) < 3e-2
    assert relm(torch.cat([p[1] for p in plan]).numpy(), ga["values"]) < 3e-2 and relm(torch.cat([p[2] for p in plan]).numpy(), ga["c_values"]) < 3e-2
    # sampler_select keeps rows -> new caches -> new plans; the step still runs
    m.sampler_select([0, 2])
    with torch.no_grad():
        o, _ = m({k: v[0:1, [0, 2]] for k, v in obs.items()}, None, pa[0:1, [0, 2]], mk[0:1, [0, 2]])
    assert o.values.shape == (1, 2, 1) and torch.isfinite(o.distributions.logits).all()
    # train mode: device-resident seed advances per step -> same inputs, different outputs
    m.train()
    for t in m.towers:
        t.time_step_counter, t._kv = 0, None
    with torch.no_grad():
        a, _ = m({k: v[0:1] for k, v in obs.items()}, None, pa[0:1], mk[0:1])
        for t in m.towers:
            t.time_step_counter = 0
        b, _ = m({k: v[0:1] for k, v in obs.items()}, None, pa[0:1], mk[0:1])
    assert (a.values - b.values).abs().max() > 1e-3 and torch.isfinite(b.values).all()


def test_recorded_small_update_equals_eager_update(ops):
    """Small minibatches: the first epoch of an update records every env-chunk's launch sequence (three ops.LaunchPlans per chunk), the
    other epochs replay them.  Same kernels on the same buffers => the update must equal the eagerly issued one (eval mode: no noise) up to
    the order of the fp32 weight-gradient atomics; in train mode every replayed epoch draws fresh dropout noise (device-resident seed)."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    torch.manual_seed(0)
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV).eval()
    st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=6, B=4, L=5, task="PickUp", seed=3), device=DEV)
    ar = m.arena
    p0 = ar.flat_p.clone()
    res = {}
    for rec in (False, True):
        ar.flat_p.copy_(p0); ar.flat_m.zero_(); ar.flat_v.zero_()
        m.sync_weights(frozen=False)
        eng = PPOLagEngine(m, PPOLagConfig(update_repeats=3, env_chunk=2, record_small_updates=rec))
        info = eng.update(st, nxt["next_value"], nxt["next_c_value"], 3.0, 1.0)
        res[rec] = (info, ar.flat_p.clone(), len(eng._chunk_cache))
    (ia, pa, na), (ib, pb, nb) = res[False], res[True]
    assert na == 0 and nb == 2                                     # two env-chunks recorded
    for k in ("value", "action", "entropy", "c_value"):
        assert abs(ia[k] - ib[k]) <= 2e-3 * max(1.0, abs(ia[k])), (k, ia[k], ib[k])
    moved = (pa - p0).norm().item()
    assert moved > 0 and (pa - pb).norm().item() < 0.05 * moved, ((pa - pb).norm().item(), moved)
    # train mode: replays run and each epoch sees different noise (the loss sums of two identical replays differ)
    m.train()
    eng = PPOLagEngine(m, PPOLagConfig(update_repeats=1, record_small_updates=True))
    batch = st.batch_slice(0, 4)
    outs = []
    for _ in range(3):
        m.zero_grad(); eng._sums.zero_()
        eng._accumulate(batch, 24, 0.1, cache_key="k")
        outs.append((eng._sums.clone(), ar.flat_g.norm().item()))
    assert len(eng._chunk_cache) == 1 and all(np.isfinite(g) and g > 0 for _, g in outs)
    assert not torch.equal(outs[1][0], outs[2][0])                 # both are replays: same launches, fresh seeds
    ar.flat_p.copy_(p0)
    m.sync_weights(frozen=False)


def test_goal_row_hash_identity_and_separation(ops):
    """svla_row_hash_u8 de-duplicates the 1000-byte goal rows (allenact_dino_transformer.py:591-603 tokenises every row): equal rows must
    hash equal; rows that differ in one byte, in byte order, or only in where the text sits inside the zero padding must hash apart; no
    collisions among 200 k random rows."""
    from safevla_amd.text import str_to_bytes

    base = np.stack([str_to_bytes(s).reshape(-1) for s in ("find a mug", "find a mug", "find a mua", "find a gum", " find a mug", "find a mug ")])
    h = ops.row_hash(torch.from_numpy(base).to(DEV)).cpu().numpy()
    assert h[0] == h[1] and len({int(x) for x in h[[0, 2, 3, 4, 5]]}) == 5 and (h >= 0).all()
    rs = np.random.RandomState(0)
    rows = rs.randint(0, 256, size=(200_000, 64), dtype=np.uint8)
    rows[1] = rows[0]
    hh = ops.row_hash(torch.from_numpy(rows).to(DEV)).cpu().numpy()
    assert hh[0] == hh[1] and len(np.unique(hh)) == len(np.unique(rows, axis=0))
    # int64 token-id rows (the synthetic path hashes the ids' bytes)
    ids = torch.from_numpy(rs.randint(3, 32000, size=(512, 12))).to(DEV)
    ids[7] = ids[3]
    hi = ops.row_hash(ids.view(torch.uint8).view(512, -1)).cpu().numpy()
    assert hi[7] == hi[3] and len(np.unique(hi)) == 511


def test_attention_over_the_full_kv_window_behind_the_mask(ops):
    """The recorded acting step attends over the WHOLE cache window (S = max_steps = 500, one query per env, kv_rows = 500) with the
    episode-window key mask: must equal attention over just the valid slots -- including an env whose window is a single slot."""
    rows, H, cap = 5, 8, 500
    cache = rnd(rows * cap, 2 * H * 64, seed=1).to(torch.bfloat16)
    q1 = rnd(rows, 3 * H * 64, seed=2).to(torch.bfloat16)
    t_now = 137
    start = torch.tensor([0, 100, 137, 50, 136])                      # per-env first valid slot (episode start), inclusive up to t_now
    kvalid = ((torch.arange(cap)[None] <= t_now) & (torch.arange(cap)[None] >= start[:, None])).to(torch.uint8)
    out, _ = ops.attn_fwd(q1.to(DEV), cache.to(DEV), cache.to(DEV)[:, H * 64:], 2 * H * 64, rows, cap, H, 0.125, kvalid=kvalid.to(DEV),
                          save_lse=False, Sq=1, ldq=3 * H * 64, kv_rows=cap)
    kk = cache.float().view(rows, cap, 2, H, 64)
    s = torch.einsum("rhd,rkhd->rhk", q1.float()[:, :H * 64].view(rows, H, 64), kk[:, :, 0]) * 0.125
    s = s.masked_fill(~kvalid.bool()[:, None, :], float("-inf"))
    want = torch.einsum("rhk,rkhd->rhd", torch.softmax(s, -1), kk[:, :, 1])
    err = (out.float().cpu().view(rows, H, 64) - want).abs().max().item() / want.abs().max().item()
    assert err < 1e-2, err
    assert torch.allclose(out.float().cpu().view(rows, H, 64)[2], kk[2, t_now, 1], atol=2e-2)      # single-slot window returns that V row


def test_t5_dropout_per_row_switch(ops):
    """Train-mode T5 dropout: one realisation per unique goal by default (rows of one episode share their text features), one per (t, b)
    row with ``t5_dropout_per_row`` (what the reference does by re-encoding every row); eval mode: identical either way."""
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)
    T, B = 4, 2
    obs, pa, mk = _obs(T, B, seed=5)                     # every step of an env carries the same goal ids

    def text_feats(per_row, train):
        m.train(train)
        m.t5_dropout_per_row = per_row
        prep = m.prepare(obs, pa, mk)
        feats = m.visual_encoder.text_encoder.encode(prep.ids, prep.attn_mask_u8, drop_seed=123 if train else None).view(prep.U, prep.L, 512)
        return prep, feats[prep.gid.long()].float()      # per (t*B + b) row

    p0, f0 = text_feats(False, True)
    assert p0.U == B and torch.equal(f0[0], f0[B])       # same env, steps 0 and 1: shared realisation
    p1, f1 = text_feats(True, True)
    assert p1.U == T * B and not torch.equal(f1[0], f1[B])
    _, e0 = text_feats(False, False)
    _, e1 = text_feats(True, False)                      # eval: the switch is inert (de-duplicated) and features agree
    assert torch.equal(e0, e1)
    m.t5_dropout_per_row = False


def test_launch_plan_single_call_replay_equals_call_by_call(ops):
    """ops.LaunchPlan.replay() re-issues a recorded sequence through ONE foreign call (svla_replay_calls, whose dispatcher is generated
    from include/svla.h): pointers / ints by value, floats and doubles by bit pattern, NULLs, the dropout descriptor.  Same results as the
    call-by-call replay and as the recorded execution itself."""
    torch.manual_seed(3)
    M, N, K = 300, 512, 256
    A = torch.randn(M, K, device=DEV).bfloat16(); W = (torch.randn(N, K, device=DEV) * 0.1).bfloat16(); bias = torch.randn(N, device=DEV)
    res = torch.randn(M, N, device=DEV).bfloat16(); g = torch.rand(N, device=DEV) + 0.5; b = torch.randn(N, device=DEV)
    y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); z = torch.empty_like(y)
    drop = ops.Dropout(seed=11, stream=4, p=0.1)
    plan = ops.LaunchPlan()
    with plan:
        ops.gemm_nt(A, W, M, N, K, bias=bias, residual=res, out=y, alpha=0.37, drop=drop)
        ops.norm_fwd(y, g, b, 1e-5, M, y=z)
        ops.dropout_(z, ops.Dropout(seed=5, stream=1, p=0.25))
    torch.cuda.synchronize()
    want = z.clone()
    assert len(plan.calls) == 3
    for replay in (plan.replay, plan.replay_python, plan.replay):
        y.zero_(); z.zero_()
        replay()
        torch.cuda.synchronize()
        assert torch.equal(z, want)
