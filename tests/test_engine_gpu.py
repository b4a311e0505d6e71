"""PPO-Lagrangian update engine on the GPU: storage + fused GAE, env-chunked gradient accumulation, lambda update,
fused clip+Adam over the flat arena -- against the CPU oracle on the reference-generated observation block."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.detfill import fill_state_dict
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)
    fill_state_dict(m, seed=7)
    m.sync_weights()
    m.eval()        # chunk-accumulation exactness / oracle comparisons need the (row-indexed) dropout noise off
    g = dict(np.load(os.path.join(G, "g5_samelen.npz"), allow_pickle=False))
    return m, g


def _storage_from_fixture(g, model):
    from safevla_amd.storage import RolloutStorage

    obs = {k[4:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("obs:")}
    T, B = g["prev_actions"].shape
    st = RolloutStorage(T - 1, device=DEV)          # use steps 0..T-2 as the rollout, step T-1 as the bootstrap observation
    st.initialize({k: v[0] for k, v in obs.items()}, num_samplers=B)
    rs = np.random.RandomState(5)
    for t in range(T - 1):
        st.add({k: v[t + 1] for k, v in obs.items()}, None, torch.from_numpy(g["prev_actions"][t + 1]).to(DEV),
               torch.from_numpy(g["batch:old_action_log_probs"][t]).to(DEV), torch.from_numpy(g["batch:values"][t]).to(DEV),
               torch.from_numpy(rs.standard_normal((B, 1)).astype(np.float32)).to(DEV),
               torch.from_numpy(rs.binomial(5, 0.2, (B, 1)).astype(np.float32)).to(DEV),
               torch.from_numpy(g["batch:c_returns"][t]).to(DEV), torch.from_numpy(g["masks"][t + 1]).to(DEV))
    st.actions.copy_(torch.from_numpy(g["batch:actions"][: T - 1]).to(DEV))
    return st, obs


def test_storage_roundtrip_and_gae(setup):
    from oracle.ref_rollout import gae_scan

    model, g = setup
    st, obs = _storage_from_fixture(g, model)
    T, B = st.T, st.B
    # tokens stored == feat_to_tokens of the raw features
    want = obs["rgb_dinov2"][: T + 1].flatten(3).permute(0, 1, 3, 2).to(torch.bfloat16)
    assert torch.equal(st.observations["dino_tokens"][:, :, 0], want)
    nv, ncv = torch.randn(B, 1, device=DEV), torch.randn(B, 1, device=DEV)
    st.compute_returns(nv, ncv, True, 0.99, 0.95)
    ret, adv = gae_scan(st.rewards.cpu(), st.value_preds[:T].cpu(), st.masks.cpu(), nv.cpu())
    cret, cadv = gae_scan(st.costs.cpu(), st.c_value_preds[:T].cpu(), st.masks.cpu(), ncv.cpu())
    assert torch.equal(st.returns[:T].cpu(), ret) and torch.equal(st.adv_targ.cpu(), adv)
    assert torch.equal(st.c_returns[:T].cpu(), cret) and torch.equal(st.c_adv_targ.cpu(), cadv)
    nxt = st.agent_input_for_next_step()
    assert nxt["prev_actions"].shape == (1, B) and nxt["masks"].shape == (1, B, 1)
    st.after_updates()
    assert st.step == 0


def test_chunked_accumulation_is_exact_and_matches_oracle(setup):
    from oracle import ref_loss, ref_model
    from oracle.detfill import grad_probe
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.text import GoalTokenizer

    model, g = setup
    st, obs = _storage_from_fixture(g, model)
    T, B = st.T, st.B
    st.compute_returns(torch.zeros(B, 1, device=DEV), torch.zeros(B, 1, device=DEV), True, 0.99, 0.95)
    lam = 0.37
    grads = {}
    for chunk in (None, 1):
        eng = PPOLagEngine(model, PPOLagConfig(env_chunk=chunk))
        model.zero_grad()
        eng._sums.zero_()
        for c0 in range(0, B, chunk or B):
            eng._accumulate(st.batch_slice(c0, min(B, c0 + (chunk or B))), T * B, lam)
        grads[chunk] = model.arena.flat_g.clone()
        sums = eng._sums.cpu().numpy() / (T * B)
    a, b = grads[None], grads[1]
    assert (a - b).norm() <= 2e-3 * a.norm(), ((a - b).norm() / a.norm()).item()   # atomics order / bf16 re-rounding only
    # oracle on the same batch
    ref = ref_model.RefSafeActorCritic(GoalTokenizer(), max_batch=B).eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()})
    bt = st.batch_slice(0, B)
    robs = {k: v[:T].cpu() for k, v in obs.items()}
    out, _ = ref(robs, None, bt["prev_actions"].cpu(), bt["masks"].cpu())
    cb = {k: bt[k].cpu() for k in ("actions", "old_action_log_probs", "adv_targ", "c_adv_targ", "returns", "values", "c_returns")}
    total, info = ref_loss.safe_ppo_log_grad(out["logits"], out["values"], cb, lam)
    c_loss = ref_loss.safe_ppo_value(out["c_values"], cb["c_returns"])
    (total + c_loss).backward()
    np.testing.assert_allclose([0.5 * sums[0], sums[1], sums[2], 0.5 * sums[4]], [info["value"], info["action"], info["entropy"], c_loss.item()], rtol=3e-2, atol=3e-2)
    named = dict(model.named_parameters())
    rn = dict(ref.named_parameters())
    errs = []
    for n, p in named.items():
        if rn[n].grad is None or float(rn[n].grad.abs().sum()) == 0:
            continue
        nrm, prj = grad_probe(n, p.grad)
        wn, wp = grad_probe(n, rn[n].grad)
        errs.append((max(abs(nrm - wn) / wn, abs(prj - wp) / wn), n))
    errs.sort()
    # bf16 activations vs the fp32 oracle on a 15-row batch: median ~1e-2, tail < 1.2e-1 (tolerance ladder)
    assert errs[len(errs) // 2][0] < 2.5e-2 and errs[-1][0] < 0.12, (errs[len(errs) // 2], errs[-3:])


def test_full_update_lambda_adam_clip(setup):
    from oracle.ref_rollout import RefLagrange
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine

    model, g = setup
    st, _ = _storage_from_fixture(g, model)
    B = st.B
    cfg = PPOLagConfig(update_repeats=2, cost_limit=2.0, lr=2e-5)
    eng = PPOLagEngine(model, cfg)
    p0 = model.arena.flat_p.clone()
    info = eng.update(st, torch.zeros(B, 1, device=DEV), torch.zeros(B, 1, device=DEV), episode_cost_sum=30.0, n_episodes=6.0)
    ref = RefLagrange(2.0, 0.001, 0.035)
    assert abs(info["lagrangian_multiplier"] - ref.update(5.0)) < 1e-6 and info["Jc"] == 5.0
    assert eng.opt_step == 2 and info["env_steps"] == st.T * B
    assert all(np.isfinite(v) for v in info.values())
    d = (model.arena.flat_p - p0).abs()
    assert d.max().item() <= 2 * 2e-5 * 1.01 and d.max().item() > 0       # |Adam step| <= lr per step
    # bf16 mirror and transposes are in sync with the fp32 masters
    assert torch.equal(model.arena.flat_bf16, model.arena.flat_p.to(torch.bfloat16))
    w = model.visual_encoder.fusion_xformer.layers[0].linear1.weight
    assert torch.equal(model._wt["f0.l1"], w.detach().to(torch.bfloat16).t().contiguous())
    # second update keeps lambda state moving in the same direction (Jc > limit)
    info2 = eng.update(st, torch.zeros(B, 1, device=DEV), torch.zeros(B, 1, device=DEV), episode_cost_sum=30.0, n_episodes=6.0)
    assert info2["lagrangian_multiplier"] > info["lagrangian_multiplier"]
    # more minibatches than environments: refused with a message before anything is touched, not a 0 / 0 in the loss bookkeeping
    with pytest.raises(ValueError, match="without environments"):
        e2 = PPOLagEngine(model, PPOLagConfig(update_repeats=1, num_mini_batch=B + 1))
        lam0 = e2.lagrange.lagrangian_multiplier
        try:
            e2.update(st, torch.zeros(B, 1, device=DEV), torch.zeros(B, 1, device=DEV), 30.0, 6.0)
        finally:
            assert e2.lagrange.lagrangian_multiplier == lam0 and e2.opt_step == 0
    # restore weights for other tests
    model.arena.flat_p.copy_(p0)
    model.sync_weights(frozen=False)


def test_checkpoint_roundtrip_and_il_handoff(setup, tmp_path):
    from safevla_amd.checkpoint import init_towers_from_il, load_checkpoint, save_checkpoint
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine

    model, g = setup
    eng = PPOLagEngine(model, PPOLagConfig())
    eng.opt_step = 7
    p0 = model.arena.flat_p.clone()
    path = str(tmp_path / "ck.pt")
    save_checkpoint(path, model, eng, total_steps=1234)
    model.arena.flat_p.mul_(0.5)
    eng.opt_step = 0
    load_checkpoint(path, model, eng)
    assert torch.equal(model.arena.flat_p, p0) and eng.opt_step == 7
    assert torch.equal(model.arena.flat_bf16, p0.to(torch.bfloat16))           # mirrors refreshed by load_state_dict
    # Lightning IL checkpoint: "model." prefix, actor.weight -> actor.linear.weight, loaded into every tower
    il = {"state_dict": {"model.actor.weight": torch.full((20, 512), 0.25), "model.actor.bias": torch.full((20,), -1.0),
                         "model.decoder.norm.weight": torch.full((512,), 3.0), "model.visual_encoder.image_encoder.model.x": torch.zeros(1)}}
    init_towers_from_il(model, il)
    for t in model.towers:
        assert (t.actor.linear.weight == 0.25).all() and (t.actor.linear.bias == -1).all() and (t.decoder.norm.weight == 3).all()
    model.arena.flat_p.copy_(p0)
    model.sync_weights(frozen=False)


def test_train_entry_point_smoke(tmp_path):
    from safevla_amd import train

    rc = train.main(["train", "--num_train_processes", "2", "--num_steps", "8", "--total_steps", "16", "--cost_limit", "2.31964",
                     "--output_dir", str(tmp_path), "--tag", "smoke", "--save_interval", "16"])
    assert rc == 0
    assert any(f.endswith(".pt") for f in os.listdir(tmp_path))
