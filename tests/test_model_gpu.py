"""End-to-end parity of the HIP three-tower actor-critic + fused losses against outputs of the REFERENCE ITSELF
(tests/golden/g5_*.npz) and against the fp32 CPU oracle, on identical name-seeded weights and seeded inputs.

Tolerance ladder (documented in DESIGN.md): the reference computes in fp32; the MI355X path stores activations and
GEMM operands in bf16 (fp32 accumulate, fp32 statistics / heads / losses).  Through 3 fusion + 3 decoder layers that
gives ~1e-2 relative error on outputs and on gradient checksums; fp32-only kernels (GAE, loss, Adam) are checked at
1e-5..bit-exact in test_kernels_gpu.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def _load(name):
    return dict(np.load(os.path.join(G, name), allow_pickle=False))


@pytest.fixture(scope="module")
def model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.detfill import fill_state_dict
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate

    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)
    fill_state_dict(m, seed=7)
    m.sync_weights()
    m.eval()        # golden vectors are eval-mode (the reference's train-mode dropout is covered by test_train_mode_dropout_vs_oracle)
    return m


def test_state_dict_names_match_reference(model):
    want = [l.rstrip("\n").split("\t") for l in open(os.path.join(G, "state_dict_manifest.txt"))]
    have = {k: str(tuple(v.shape)) for k, v in model.state_dict().items()}
    assert set(have) == {k for k, _ in want}
    for k, shp in want:
        assert have[k] == shp, (k, have[k], shp)


def test_t5_encoder_vs_oracle(model):
    from oracle.detfill import fill_state_dict
    from oracle.ref_t5 import RefT5Encoder

    from safevla_amd.model import T5Frozen

    # own instance: name-seeded weights with the query projection scaled down so the (un-scaled, T5-style) softmax is
    # not saturated -- a saturated softmax turns bf16 rounding into arg-max flips, which tests chaos, not the kernels
    t5 = T5Frozen(torch.device(DEV))
    fill_state_dict(t5, seed=11)
    with torch.no_grad():
        for b in t5.encoder.block:
            b.layer[0].SelfAttention.q.weight.mul_(0.25)
    t5.sync()
    ref = RefT5Encoder().eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in t5.state_dict().items()})
    rs = np.random.RandomState(0)
    ids = torch.from_numpy(rs.randint(3, 32000, size=(5, 11)))
    am = torch.ones(5, 11, dtype=torch.int64)
    for i, n in enumerate([11, 4, 7, 1, 9]):
        ids[i, n:] = 0
        am[i, n:] = 0
    want = ref(ids, am)
    got = t5.encode(ids.to(DEV), am.to(DEV)).float().view(5, 11, 512).cpu()
    valid = am.bool()
    err = (got - want)[valid].abs().max().item()
    assert err < 0.04 * want[valid].abs().max().item(), (err, want[valid].abs().max().item())


@pytest.mark.parametrize("big_tiles", [False, True])
@pytest.mark.parametrize("prune_last", [True, False])
@pytest.mark.parametrize("tag", ["g5_samelen", "g5_mixedlen"])
def test_three_towers_forward_backward_vs_reference(model, tag, prune_last, big_tiles):
    """prune_last: the last fusion layer computed only for the consumed sequence position 0 (default) vs. in full --
    both must reproduce the reference's outputs and gradients.  big_tiles: at the golden's size (32 rows x 177 tokens) the GEMM
    dispatch would pick the 128-tile kernels; forcing the 256-tile ones (gemm_nt8p / gemm_tn8p, the kernels that carry 70 % of an
    update) runs the reference's own fixtures through them (the attention kernels are chosen by S, not by the row count)."""
    from oracle.detfill import grad_probe
    from safevla_amd import ops
    from safevla_amd.losses import SafePPOLogGrad, SafePPOValue

    if big_tiles and not prune_last:
        pytest.skip("one pruning mode is enough for the kernel-selection variant")
    ops.gemm_force_small_tile(2 if big_tiles else 0)
    try:
        _three_towers_vs_reference(model, tag, prune_last, grad_probe, SafePPOLogGrad, SafePPOValue)
    finally:
        ops.gemm_force_small_tile(0)


def _three_towers_vs_reference(model, tag, prune_last, grad_probe, SafePPOLogGrad, SafePPOValue):
    for t in model.towers:
        t.prune_last = prune_last
    g = _load(tag + ".npz")
    obs = {k[4:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("obs:")}
    batch = {k[6:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("batch:")}
    model.zero_grad()
    aco, _ = model(obs, None, torch.from_numpy(g["prev_actions"]).to(DEV), torch.from_numpy(g["masks"]).to(DEV))
    lg = aco.distributions.logits.detach().float().cpu().numpy()
    # ---- forward vs the reference's own outputs
    def rel(a, b):
        return np.abs(a - b).max() / (np.abs(b).max() + 1e-12)

    e_l, e_v, e_c = rel(lg, g["logits"]), rel(aco.values.detach().cpu().numpy(), g["values"]), rel(aco.c_values.detach().cpu().numpy(), g["c_values"])
    print(f"[{tag}] rel-to-max err: logits {e_l:.3e} values {e_v:.3e} c_values {e_c:.3e}")
    # gate = 2x the largest error measured over every kernel-selection variant (4e-3 ... 1.5e-2); VERDICT r5 item 6 (was 3e-2)
    assert e_l < 2e-2 and e_v < 2e-2 and e_c < 2e-2
    # ---- losses (fused HIP) vs the reference's SafePPOLogGrad scalars
    loss = SafePPOLogGrad(clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.0, use_clipped_value_loss=False,
                          action_loss_schedule=None, discrete_critics=False, normalize_advantage=False)
    total, info = loss.loss(0, batch, aco, lagrangian_multiplier=torch.tensor(float(g["lam"])))
    c_total, c_info = SafePPOValue(clip_param=0.1, use_clipped_value_loss=False).loss(0, batch, aco)
    for k in ("ppo_total", "value", "action", "entropy"):
        assert abs(info[k] - float(g[k])) < 3e-2 * max(1.0, abs(float(g[k]))), (k, info[k], float(g[k]))
    assert abs(c_info["c_value"] - float(g["c_value_loss"])) < 3e-2 * max(1.0, float(g["c_value_loss"]))
    (total + c_total).backward()
    # ---- gradient checksums of all 252 trainable tensors the reference gives gradients to
    named = dict(model.named_parameters())
    worst = []
    for n in g["grad_names"]:
        n = str(n)
        nrm, prj = grad_probe(n, named[n].grad)
        wn, wp = g["gp:" + n]
        worst.append((abs(nrm - wn) / (wn + 1e-12), abs(prj - wp) / (wn + 1e-12), n))
    worst.sort(reverse=True)
    print(f"[{tag}] worst grad-norm rel errs:", [(f"{a:.2e}", f"{b:.2e}", n) for a, b, n in worst[:5]])
    bad = [w for w in worst if w[0] > 4e-2 or w[1] > 4e-2]      # measured <= 3.2e-2; was 6e-2
    assert not bad, bad[:10]
    # exactly the reference's set of parameters receives gradient
    have = {n for n, p in named.items() if p.grad is not None and float(p.grad.abs().sum()) > 0}
    assert have == {str(n) for n in g["grad_names"]}, have ^ {str(n) for n in g["grad_names"]}
    for t in model.towers:
        t.prune_last = True


def test_forward_is_deterministic_and_no_grad_path(model):
    g = _load("g5_samelen.npz")
    obs = {k[4:]: torch.from_numpy(v).to(DEV) for k, v in g.items() if k.startswith("obs:")}
    pa, mk = torch.from_numpy(g["prev_actions"]).to(DEV), torch.from_numpy(g["masks"]).to(DEV)
    with torch.no_grad():
        a, _ = model(obs, None, pa, mk)
        b, _ = model(obs, None, pa, mk)
    assert torch.equal(a.distributions.raw_logits, b.distributions.raw_logits)
    assert torch.equal(a.values, b.values) and torch.equal(a.c_values, b.c_values)


def test_acting_path_kv_cache_vs_reference(model):
    """nsteps == 1 with the llama KV cache and the episode-window mask (reference acting path) vs the reference's own
    step-by-step outputs, and vs this model's full-sequence update path (equal for equal token lengths)."""
    g, gu = _load("g5_acting.npz"), _load("g5_samelen.npz")
    obs = {k[4:]: torch.from_numpy(v).to(DEV) for k, v in gu.items() if k.startswith("obs:")}
    pa, mk = torch.from_numpy(gu["prev_actions"]).to(DEV), torch.from_numpy(gu["masks"]).to(DEV)
    T = pa.shape[0]
    for t in model.towers:
        t.time_step_counter = 0
        t._kv = None
    lg, vs, cs = [], [], []
    with torch.no_grad():
        for t in range(T):
            o, _ = model({k: v[t:t + 1] for k, v in obs.items()}, None, pa[t:t + 1], mk[t:t + 1])
            lg.append(o.distributions.logits); vs.append(o.values); cs.append(o.c_values)
    lg, vs, cs = torch.cat(lg).cpu().numpy(), torch.cat(vs).cpu().numpy(), torch.cat(cs).cpu().numpy()
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    assert rel(lg, g["logits"]) < 3e-2 and rel(vs, g["values"]) < 3e-2 and rel(cs, g["c_values"]) < 3e-2, (rel(lg, g["logits"]), rel(vs, g["values"]))
    assert all(t.time_step_counter == T for t in model.towers)
    with torch.no_grad():
        full, _ = model(obs, None, pa, mk)            # T > 1 resets the counters
    assert all(t.time_step_counter == 0 for t in model.towers)
    assert rel(lg, full.distributions.logits.cpu().numpy()) < 3e-2
    model.sampler_select([0, 2])
    assert model._kv[0].shape[0] == 2
    for t in model.towers:
        t._kv = None


@pytest.mark.parametrize("prune_last", [True, False])
def test_train_mode_dropout_vs_oracle(model, prune_last):
    """The reference keeps the policy in train() mode (allenact_dino_transformer.py:193): dropout 0.1 on the attention
    probabilities, both sub-layer outputs and the feed-forward activation of every fusion layer.  The HIP path regenerates its
    keep-masks from element indices (include/svla.h: svla_dropout); the oracle applies the SAME masks (oracle.ref_model.hash_dropout),
    so train-mode outputs and gradients must agree like the eval-mode ones do -- forward, backward, pruned and full last layer."""
    from oracle import ref_loss, ref_model
    from oracle.detfill import fill_state_dict, grad_probe
    from safevla_amd.losses import SafePPOLogGrad, SafePPOValue
    from safevla_amd.text import GoalTokenizer

    g = _load("g5_samelen.npz")
    T, B = 3, 2                                   # a slice of the fixture keeps the fp32 CPU oracle at seconds
    cut = lambda v: v[:T, :B]
    obs_np = {k[4:]: cut(v) for k, v in g.items() if k.startswith("obs:")}
    batch_np = {k[6:]: cut(v) for k, v in g.items() if k.startswith("batch:")}
    pa_np, mk_np = cut(g["prev_actions"]), cut(g["masks"])
    model.train()
    seeds = []
    for k, t in enumerate(model.towers):
        t.prune_last = prune_last
        t.drop_seed_base, t._fwd_count = 1000 + k, 0
        t.t5_dropout = False      # text-encoder noise is compared separately (test_t5_train_mode_dropout_vs_oracle)
        seeds.append(((1000 + k) * 0x9E3779B1 + 1 * 0x85EBCA77) & 0xFFFFFFFF)
    try:
        model.zero_grad()
        aco, _ = model({k: torch.from_numpy(v).to(DEV) for k, v in obs_np.items()}, None, torch.from_numpy(pa_np).to(DEV), torch.from_numpy(mk_np).to(DEV))
        batch = {k: torch.from_numpy(v).to(DEV) for k, v in batch_np.items()}
        loss = SafePPOLogGrad(clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.0, use_clipped_value_loss=False,
                              action_loss_schedule=None, discrete_critics=False, normalize_advantage=False)
        total, info = loss.loss(0, batch, aco, lagrangian_multiplier=torch.tensor(0.37))
        c_total, _ = SafePPOValue(clip_param=0.1, use_clipped_value_loss=False).loss(0, batch, aco)
        (total + c_total).backward()
    finally:
        model.eval()
        for t in model.towers:
            t.prune_last = True
            t.t5_dropout = True
    # ---- oracle, train mode, same masks
    ref = ref_model.RefSafeActorCritic(GoalTokenizer(), max_batch=B, dropout=0.1)
    fill_state_dict(ref, seed=7)
    ref.train()
    for tower, seed in zip([ref, ref.critic_tsfm, ref.c_critic_tsfm], seeds):
        tower.visual_encoder.text_encoder.eval()          # T5 noise is not part of this comparison
        for l in tower.visual_encoder.fusion_xformer.layers:
            l.hash_seed = seed
    out, _ = ref({k: torch.from_numpy(v) for k, v in obs_np.items()}, None, torch.from_numpy(pa_np), torch.from_numpy(mk_np))
    rel = lambda a, b: np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
    e_l = rel(aco.distributions.logits.detach().float().cpu().numpy(), torch.log_softmax(out["logits"], -1).detach().numpy())
    e_v = rel(aco.values.detach().cpu().numpy(), out["values"].detach().numpy())
    print(f"[dropout prune_last={prune_last}] rel-to-max err: logits {e_l:.3e} values {e_v:.3e}")
    assert e_l < 3e-2 and e_v < 3e-2, (e_l, e_v)
    assert rel(aco.c_values.detach().cpu().numpy(), out["c_values"].detach().numpy()) < 3e-2
    # eval-mode outputs differ clearly: the masks really are applied
    with torch.no_grad():
        ev, _ = model({k: torch.from_numpy(v).to(DEV) for k, v in obs_np.items()}, None, torch.from_numpy(pa_np).to(DEV), torch.from_numpy(mk_np).to(DEV))
    assert rel(ev.values.cpu().numpy(), out["values"].detach().numpy()) > 5e-2
    rb = {k: torch.from_numpy(v) for k, v in batch_np.items()}
    r_total, _ = ref_loss.safe_ppo_log_grad(out["logits"], out["values"], rb, 0.37, clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.0,
                                            use_clipped_value_loss=False)
    r_c = ref_loss.safe_ppo_value(out["c_values"], rb["c_returns"])
    (r_total + r_c).backward()
    named, rnamed = dict(model.named_parameters()), dict(ref.named_parameters())
    errs = []
    for n in g["grad_names"]:
        n = str(n)
        nrm, prj = grad_probe(n, named[n].grad)
        wn, wp = grad_probe(n, rnamed[n].grad)
        errs.append((abs(nrm - wn) / (wn + 1e-12), n))
    errs.sort(reverse=True)
    assert np.median([e for e, _ in errs]) < 2e-2 and errs[0][0] < 1e-1, errs[:5]


def test_t5_train_mode_dropout_vs_oracle(model):
    """The frozen T5 encoder stays in train mode with the rest of the policy: its dropout sites (embedding,
    attention probabilities, both residual branches, feed-forward activation, final norm) with the shared counter-based masks."""
    from oracle.detfill import fill_state_dict
    from oracle.ref_t5 import RefT5Encoder
    from safevla_amd.model import T5Frozen

    t5 = T5Frozen(torch.device(DEV))
    fill_state_dict(t5, seed=11)
    with torch.no_grad():
        for b in t5.encoder.block:
            b.layer[0].SelfAttention.q.weight.mul_(0.25)     # un-saturated softmax, as in test_t5_encoder_vs_oracle
    t5.sync()
    ref = RefT5Encoder().train()
    ref.load_state_dict({k: v.detach().cpu() for k, v in t5.state_dict().items()})
    ref.hash_seed = 4242
    rs = np.random.RandomState(3)
    U, L = 5, 11
    ids = torch.from_numpy(rs.randint(3, 32000, size=(U, L)).astype(np.int64))
    am = torch.ones(U, L, dtype=torch.int64)
    for u, n in enumerate([11, 4, 7, 9, 6]):
        ids[u, n - 1] = 1; ids[u, n:] = 0; am[u, n:] = 0
    got = t5.encode(ids.to(DEV), am.to(DEV), drop_seed=4242).float().view(U, L, 512).cpu().numpy()
    want = ref(ids, am).numpy()
    plain = t5.encode(ids.to(DEV), am.to(DEV)).float().view(U, L, 512).cpu().numpy()
    valid = am.numpy().astype(bool)
    err = np.abs(got - want)[valid].max() / np.abs(want[valid]).max()
    assert err < 3e-2, err
    assert np.abs(plain - want)[valid].max() / np.abs(want[valid]).max() > 0.2      # dropout really changes the features
    assert np.mean(got[valid] == 0) > 0.08                                          # final-site zeros (p = 0.1)


def test_acting_graph_replay_equals_eager_acting(model):
    """The captured single-step graph (device-resident step counter / KV slot / dropout seed, attention over the whole cache window
    behind the mask) must reproduce the eager acting path step for step, across an episode boundary, and keep running in train mode."""
    gu = _load("g5_samelen.npz")
    obs = {k[4:]: torch.from_numpy(v).to(DEV) for k, v in gu.items() if k.startswith("obs:")}
    pa, mk = torch.from_numpy(gu["prev_actions"]).to(DEV), torch.from_numpy(gu["masks"]).to(DEV)
    T = pa.shape[0]

    def run(graph):
        for t in model.towers:
            t.time_step_counter, t._kv = 0, None
        model.enable_acting_graphs(graph)
        out = []
        with torch.no_grad():
            for t in range(T):
                o, _ = model({k: v[t:t + 1] for k, v in obs.items()}, None, pa[t:t + 1], mk[t:t + 1])
                out.append((o.distributions.logits.float().cpu(), o.values.cpu(), o.c_values.cpu()))
        model.enable_acting_graphs(False)
        return out

    eager, graph = run(False), run(True)
    for t, (a, b) in enumerate(zip(eager, graph)):
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=0, atol=2e-2 * max(1.0, x.abs().max().item())), (t, (x - y).abs().max())
    assert all(t.time_step_counter == T for t in model.towers)
    # train mode: replays draw fresh dropout noise (device-resident seed) -> same inputs, different outputs
    model.train()
    try:
        for t in model.towers:
            t.time_step_counter, t._kv = 0, None
        model.enable_acting_graphs(True)
        with torch.no_grad():
            a, _ = model({k: v[0:1] for k, v in obs.items()}, None, pa[0:1], mk[0:1])
            for t in model.towers:
                t.time_step_counter = 0
            b, _ = model({k: v[0:1] for k, v in obs.items()}, None, pa[0:1], mk[0:1])
        assert (a.values - b.values).abs().max() > 1e-3
        assert torch.isfinite(a.distributions.logits).all() and torch.isfinite(b.values).all()
    finally:
        model.enable_acting_graphs(False)
        model.eval()
        for t in model.towers:
            t._kv, t.time_step_counter = None, 0


def test_fused_rmsnorm_step_close_to_unfused():
    """an acting step folds the llama decoder's (and, for small passes, the frozen T5's) RMSNorm into the following GEMM -- RMSNorm(x) W^T = rstd (x (W gamma)^T),
    no bf16 rounding of the normed row -- while the update's sequence branch runs norm -> bf16 -> GEMM.  Rollout log-probs and the update's recomputed ones therefore
    differ by that rounding; this pins how much: 12 KV-cached steps at 8 envs with the fold on and off (``rms_fused`` / ``t5_fused``), eval mode, same weights."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle.detfill import fill_state_dict
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    m = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV)
    fill_state_dict(m, seed=3)
    m.sync_weights()
    m.eval()
    B, n = 8, 12
    st, _, _ = fill_synthetic_rollout(m, SynthSpec(T=n + 1, B=B, L=12, task="PickUp", seed=9), device=DEV)
    outs = {}
    for fused in (True, False):
        for t in m.towers:
            t.time_step_counter, t._kv, t.rms_fused, t.t5_fused, t._t5_cache = 0, None, fused, fused, (None, None)
        m.invalidate_recorded()
        res = []
        with torch.no_grad():
            for t in range(n):
                o, _ = m({k: v[t:t + 1] for k, v in st.observations.items()}, None, st.prev_actions[t:t + 1], st.masks[t:t + 1])
                res.append((torch.log_softmax(o.distributions.logits.float(), -1).clone(), o.values.float().clone(), o.c_values.float().clone()))
        outs[fused] = res
    for t in m.towers:
        t.rms_fused, t.t5_fused = True, None
    scale_l = max(float(b[0].abs().max()) for b in outs[False]) + 1e-6          # (deterministic-fill weights give log-probs of order 10: relative to the largest)
    d_logp = max(float((a[0] - b[0]).abs().max()) for a, b in zip(outs[True], outs[False])) / scale_l
    scale_v = max(float(b[1].abs().max()) for b in outs[False]) + 1e-6
    d_v = max(float((a[1] - b[1]).abs().max()) for a, b in zip(outs[True], outs[False])) / scale_v
    print(f"fused vs unfused RMSNorm: max |d log p| rel-to-max {d_logp:.3e} (largest |log p| {scale_l:.2f}), values rel-to-max {d_v:.3e}")
    assert d_logp > 0.0                       # the two forms ARE different arithmetic ...
    assert d_logp < 2e-2 and d_v < 2e-2       # ... one bf16 rounding of the normed rows apart (PPO's ratio clip is 0.1)
