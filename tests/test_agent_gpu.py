"""Evaluation agent (InferenceAgentVIDA mirror): the step-by-step agent API must reproduce the update path.

The reference agent is [frames -> frozen ViT -> storage -> single-step actor-critic with KV cache -> sample].  Parity here is a
consistency chain: ViT features are checked against the fp32 oracle in test_preproc_gpu.py and the single-step model against the
reference goldens in test_model_gpu.py; this test checks that the agent wires them together exactly like one full-sequence
(update-path) forward over the same frames, previous actions, masks and time steps, across a ``reset()`` and a storage refresh."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def agent():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.detfill import fill_state_dict
    from safevla_amd.agent import InferenceAgentVIDA
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)
    fill_state_dict(m, seed=7)
    m.sync_weights()
    m.eval()        # step-by-step vs full-sequence equality needs the dropout noise off
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    return InferenceAgentVIDA.build_agent(m, device=DEV, greedy_sampling=False, steps_before_rollout_refresh=4, generator=g)


def _frames(rs, n):
    return [{"raw_navigation_camera": rs.randint(0, 256, (224, 384, 3), dtype=np.uint8),
             "raw_manipulation_camera": rs.randint(0, 256, (224, 384, 3), dtype=np.uint8),
             "an_object_is_in_hand": np.array([int(i >= 3)])} for i in range(n)]


def test_agent_api_and_action_list(agent):
    names = agent.get_action_list()
    assert len(names) == 20 and names[0] == "m" and names[4] == "end" and names[-1] == "d"
    a, probs = agent.get_action(_frames(np.random.RandomState(0), 1)[0], "find a mug")
    assert a in names and probs.shape == (20,) and abs(float(probs.sum()) - 1.0) < 1e-4
    agent.reset()


def test_agent_steps_equal_full_sequence_forward(agent):
    m = agent.actor_critic
    rs = np.random.RandomState(1)
    T = 7                                          # crosses the 4-step storage refresh
    fr = _frames(rs, T)
    agent.reset()
    traj = agent.traj_index
    acts, probs = [], []
    for t in range(T):
        a, p = agent.get_action(fr[t], "navigate to the red chair and pick up the cup")
        acts.append(int(agent.last_action_flat[0])); probs.append(p.float().cpu().numpy())
        assert a == agent.get_action_list()[acts[-1]]
    assert agent.steps_taken_in_task == T
    # the same episode as ONE update-path forward: features from the same preprocessors, shifted sampled actions, masks 0,1,1,...
    from safevla_amd.text import str_to_bytes
    u = m.uuids
    nav = torch.cat([agent.nav_pre.process({"rgb_raw": torch.from_numpy(f["raw_navigation_camera"])[None]}) for f in fr])
    man = torch.cat([agent.manip_pre.process({"manipulation_rgb_raw": torch.from_numpy(f["raw_manipulation_camera"])[None]}) for f in fr])
    goal = torch.from_numpy(np.asarray(str_to_bytes("navigate to the red chair and pick up the cup", 1000))).to(DEV).reshape(1, 1, -1).repeat(T, 1, 1)
    obs = {u["nav"]: nav[:, None], u["manip"]: man[:, None], u["goal"]: goal,
           u["time"]: torch.arange(T, device=DEV).reshape(T, 1), u["traj"]: torch.full((T, 1), traj, device=DEV, dtype=torch.int64),
           u["hand"]: torch.tensor([int(i >= 3) for i in range(T)], device=DEV).reshape(T, 1)}
    pa = torch.tensor([0] + acts[:-1], device=DEV, dtype=torch.int64).reshape(T, 1)
    mk = torch.ones(T, 1, 1, device=DEV); mk[0] = 0
    with torch.no_grad():
        full, _ = m(obs, None, pa, mk)
    ref = full.distributions.probs[:, 0].float().cpu().numpy()
    got = np.stack(probs)
    assert np.abs(got - ref).max() < 2e-2 * max(ref.max(), 1e-3), np.abs(got - ref).max()
    # a new task starts a fresh episode window: first-step output must not depend on the previous episode
    agent.reset()
    a1, p1 = agent.get_action(fr[0], "navigate to the red chair and pick up the cup")
    assert np.abs(p1.float().cpu().numpy() - ref[0]).max() < 2e-2 * max(ref[0].max(), 1e-3)


def test_agent_steps_vs_the_cpu_oracle_chain(agent):
    """Not a self-comparison: the same episode through the fp32 CPU ORACLE chain -- oracle ViT (ref_vit.vit_features) on the raw frames ->
    oracle 3-tower policy stepped with its own KV caches (ref_model, acting branch) -- with the agent's weights.  The agent's action
    probabilities must follow it on the bf16 ladder (frames -> 12 ViT blocks -> 3 fusion + 3 decoder layers)."""
    from oracle import ref_loss, ref_model, ref_vit
    from safevla_amd.text import GoalTokenizer, str_to_bytes

    m = agent.actor_critic
    rs = np.random.RandomState(2)
    T = 3
    fr = _frames(rs, T)
    goal = "find a mug"
    agent.reset()
    probs, acts = [], []
    for t in range(T):
        _, p = agent.get_action(fr[t], goal)
        probs.append(p.float().cpu().numpy()); acts.append(int(agent.last_action_flat[0]))
    ref = ref_model.RefSafeActorCritic(GoalTokenizer(), max_batch=1).eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    vsd = {k: v.detach().float().cpu() for k, v in agent.nav_pre.vit.state_dict().items()}
    gb = torch.from_numpy(np.asarray(str_to_bytes(goal, 1000))).reshape(1, 1, -1)
    want = []
    with torch.no_grad():
        for t in range(T):
            _, nav = ref_vit.vit_features(vsd, torch.from_numpy(fr[t]["raw_navigation_camera"])[None])
            _, man = ref_vit.vit_features(vsd, torch.from_numpy(fr[t]["raw_manipulation_camera"])[None])
            obs = {"rgb_dinov2": nav[None], "manipulation_rgb_dinov2": man[None], "natural_language_spec": gb,
                   "time_step": torch.tensor([[t]]), "traj_index": torch.tensor([[agent.traj_index]]),
                   "an_object_is_in_hand": torch.tensor([[[int(t >= 3)]]])}
            pa = torch.tensor([[acts[t - 1] if t else 0]])
            mk = torch.tensor([[[1.0 if t else 0.0]]])
            out, _ = ref(obs, None, pa, mk)
            want.append(torch.softmax(out["logits"][0, 0], -1).numpy())
    got, want = np.stack(probs), np.stack(want)
    err = np.abs(got - want).max() / want.max()
    assert err < 4e-2, err
    agent.reset()
