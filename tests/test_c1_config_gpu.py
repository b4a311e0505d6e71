"""BASELINE.json configs[0] (C1) as written: ObjectNav, 4 synthetic envs, 32-step rollout, one full PPO-Lagrangian update -- the
"CPU-only PyTorch reference PPO update" plumbing configuration.  The HIP path runs that exact workload (GAE, lambda update, the
production 4 epochs x 1 minibatch of 3-tower forward / fused losses / backward / clip 0.5 / Adam 2e-5) and is compared with the CPU
oracle performing the same update with torch.optim.Adam:

  * fp32 verification mode: per-epoch loss means, lambda, post-update logits / values / cost values at 1e-4 (fp32 tolerance);
  * bf16 product path: the same quantities on the documented bf16 ladder (3e-2 of max), and against the fp32 mode.

Reference for the configuration: training/online/dinov2_vits_tsfm_base.py:293-380 (update_repeats 4, num_mini_batch 1, lr 2e-5,
max_grad_norm 0.5, SafePPOLogGrad + SafePPOValue); the synthetic rollout replaces AI2-THOR (safevla_amd/synth_env.py).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
T, B, L = 32, 4, 4


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


class IdTokenizer:
    """oracle-side tokenizer of this test: the goal 'strings' are the decimal token ids themselves"""

    def __call__(self, goals, return_tensors="pt", padding=True):
        ids = torch.tensor([[int(w) for w in g.split()] for g in goals], dtype=torch.int64)
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}


@pytest.fixture(scope="module")
def c1():
    """the C1 rollout (filled through the fp32 model's own no-grad pass) + the oracle's update of it"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import ref_loss, ref_model
    from oracle.ref_rollout import RefLagrange, gae_scan
    from safevla_amd.engine import PPOLagConfig
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout
    from safevla_amd.text import str_to_bytes

    torch.manual_seed(0)
    m32 = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, precision="fp32").eval()
    sd = {k: v.detach().clone() for k, v in m32.state_dict().items()}
    st, nxt, ep = fill_synthetic_rollout(m32, SynthSpec(T=T, B=B, L=L, task="ObjectNav", seed=79, cost_p=0.3), device=DEV)
    cfg = PPOLagConfig(cost_limit=2.31964)                     # the production update: 4 epochs x 1 minibatch
    assert cfg.update_repeats == 4 and cfg.num_mini_batch == 1
    # ---- the same update on the CPU oracle
    tok = st.observations["dino_tokens"][:T].float().cpu()                      # [T, B, 2, 84, 384] (bf16-exact values)
    ids = st.observations["goal_token_ids"][:T].cpu().numpy()
    robs = {"rgb_dinov2": tok[:, :, 0].permute(0, 1, 3, 2).reshape(T, B, 384, 7, 12).contiguous(),
            "manipulation_rgb_dinov2": tok[:, :, 1].permute(0, 1, 3, 2).reshape(T, B, 384, 7, 12).contiguous(),
            "natural_language_spec": torch.from_numpy(np.stack([np.stack([str_to_bytes(" ".join(str(i) for i in ids[t, b])).reshape(-1) for b in range(B)]) for t in range(T)])),
            "time_step": st.observations["time_step"][:T].cpu(), "traj_index": st.observations["traj_index"][:T].cpu(),
            "an_object_is_in_hand": st.observations["an_object_is_in_hand"][:T].cpu()}
    ref = ref_model.RefSafeActorCritic(IdTokenizer(), max_batch=B).eval()
    ref.load_state_dict({k: v.cpu() for k, v in sd.items()})
    nv, ncv = nxt["next_value"].cpu(), nxt["next_c_value"].cpu()
    ret, adv = gae_scan(st.rewards.cpu(), st.value_preds[:T].cpu(), st.masks.cpu(), nv)
    cret, cadv = gae_scan(st.costs.cpu(), st.c_value_preds[:T].cpu(), st.masks.cpu(), ncv)
    n_ep = max(ep["n_episodes"], 1.0)
    lam = RefLagrange(cfg.cost_limit, cfg.lambda_init, cfg.lambda_lr).update(ep["episode_cost_sum"] / n_ep)
    cb = {"actions": st.actions.cpu(), "old_action_log_probs": st.action_log_probs.cpu(), "adv_targ": adv, "c_adv_targ": cadv, "returns": ret,
          "values": st.value_preds[:T].cpu(), "c_returns": cret}
    params = [p for n, p in ref.named_parameters() if "text_encoder" not in n]
    opt = torch.optim.Adam(params, lr=cfg.lr)
    pa, mk = st.prev_actions[:T].cpu(), st.masks[:T].cpu()
    acc = np.zeros(4)
    for _ in range(cfg.update_repeats):
        opt.zero_grad()
        out, _ = ref(robs, None, pa, mk)
        total, ri = ref_loss.safe_ppo_log_grad(out["logits"], out["values"], cb, lam)
        c_loss = ref_loss.safe_ppo_value(out["c_values"], cret)
        (total + c_loss).backward()
        torch.nn.utils.clip_grad_norm_(params, cfg.max_grad_norm)
        opt.step()
        acc += np.array([ri["value"], ri["action"], ri["entropy"], c_loss.item()]) / cfg.update_repeats
    with torch.no_grad():
        out, _ = ref(robs, None, pa, mk)
    post = {"logits": ref_loss.categorical(out["logits"]).numpy(), "values": out["values"].numpy(), "c_values": out["c_values"].numpy()}
    del m32
    torch.cuda.empty_cache()
    return dict(sd=sd, st=st, nxt=nxt, ep=ep, cfg=cfg, lam=lam, losses=acc, post=post)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
def test_c1_objectnav_4_envs_32_steps_full_update_vs_cpu_oracle(c1, precision, tol):
    from safevla_amd.engine import PPOLagEngine
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    torch.manual_seed(0)
    model = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, precision=precision).eval()
    model.load_state_dict(c1["sd"])
    st, nxt, ep = c1["st"], c1["nxt"], c1["ep"]
    eng = PPOLagEngine(model, c1["cfg"])
    info = eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])
    assert abs(info["lagrangian_multiplier"] - c1["lam"]) < 1e-6, (info["lagrangian_multiplier"], c1["lam"])
    assert info["lagrangian_multiplier"] != c1["cfg"].lambda_init                # the multiplier moved (cost constraint evaluated)
    got = np.array([info["value"], info["action"], info["entropy"], info["c_value"]])
    print(f"[C1 {precision}] losses gpu {got} oracle {c1['losses']}")
    np.testing.assert_allclose(got, c1["losses"], rtol=tol, atol=tol * 1e-2 if precision == "fp32" else tol)
    with torch.no_grad():
        aco, _ = model({k: v[:T] for k, v in st.observations.items()}, None, st.prev_actions[:T], st.masks[:T])
    errs = {"logits": rel(aco.distributions.logits.float().cpu().numpy(), c1["post"]["logits"]),
            "values": rel(aco.values.float().cpu().numpy(), c1["post"]["values"]),
            "c_values": rel(aco.c_values.float().cpu().numpy(), c1["post"]["c_values"])}
    print(f"[C1 {precision}] post-update rel-to-max errors {errs}")
    assert max(errs.values()) < tol, errs
    assert torch.isfinite(model.arena.flat_p).all()
