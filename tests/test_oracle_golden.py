"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz, made by make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_loss, ref_model
from oracle.detfill import fill_state_dict, grad_probe
from safevla_amd.text import GoalTokenizer

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return dict(np.load(os.path.join(G, name), allow_pickle=False))


@pytest.fixture(scope="module")
def oracle_model():
    torch.manual_seed(0)
    m = ref_model.RefSafeActorCritic(GoalTokenizer(), max_steps=500, max_batch=4).eval()
    fill_state_dict(m, seed=7)
    return m


def test_state_dict_manifest(oracle_model):
    want = [l.rstrip("\n").split("\t") for l in open(os.path.join(G, "state_dict_manifest.txt"))]
    have = {k: str(tuple(v.shape)) for k, v in oracle_model.state_dict().items()}
    assert len(want) == 417
    for k, shp in want:
        assert k in have, k
        assert have[k] == shp, (k, have[k], shp)
    assert set(have) == {k for k, _ in want}


def test_g1_hlgauss():
    g = _load("g1_hlgauss.npz")
    sup = ref_loss.hl_support()
    t, lg = torch.from_numpy(g["target"]), torch.from_numpy(g["logits"])
    np.testing.assert_allclose(ref_loss.hl_gauss_probs(t, sup).numpy(), g["probs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ref_loss.hl_gauss_loss(lg, t, sup).numpy(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose(ref_loss.hl_gauss_value(torch.softmax(lg, -1), sup).numpy(), g["value"], rtol=1e-6)


def test_g2_posenc(oracle_model):
    g = _load("g2_posenc.npz")
    np.testing.assert_allclose(oracle_model.time_encoder(torch.from_numpy(g["pos"])).numpy(), g["pe"], rtol=0, atol=1e-6)


def test_g3_decoder(oracle_model):
    g = _load("g3_decoder.npz")
    dec = oracle_model.decoder
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    traj = torch.from_numpy(g["traj"])
    mask = torch.tril(traj[:, :, None] == traj[:, None, :])[:, None]
    y = dec(x, 0, mask)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=2e-5)
    for p in dec.parameters():
        p.grad = None
    (y * torch.from_numpy(g["w"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=1e-3, atol=2e-5)
    for n, p in dec.named_parameters():
        nrm, prj = grad_probe(n, p.grad)
        np.testing.assert_allclose([nrm, prj], g["gp:" + n], rtol=2e-4, atol=1e-5)
    # KV-cache stepping == full causal sequence
    B, T = x.shape[:2]
    with torch.no_grad():
        ys = [dec(x[:, t : t + 1].detach(), t, torch.ones(B, 1, 1, t + 1, dtype=torch.bool)) for t in range(T)]
    np.testing.assert_allclose(torch.cat(ys, 1).numpy(), g["y_cache"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(g["y_cache"], g["y_causal"], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("tag", ["g5_mixedlen", "g5_samelen"])
def test_g5_three_towers(oracle_model, tag):
    g = _load(tag + ".npz")
    obs = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("obs:")}
    batch = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch:")}
    for p in oracle_model.parameters():
        p.grad = None
    out, _ = oracle_model(obs, None, torch.from_numpy(g["prev_actions"]), torch.from_numpy(g["masks"]))
    np.testing.assert_allclose(ref_loss.categorical(out["logits"]).detach().numpy(), g["logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["values"].detach().numpy(), g["values"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(out["c_values"].detach().numpy(), g["c_values"], rtol=1e-4, atol=2e-5)
    total, info = ref_loss.safe_ppo_log_grad(out["logits"], out["values"], batch, float(g["lam"]))
    for k in ("ppo_total", "value", "action", "entropy"):
        np.testing.assert_allclose(info[k], g[k], rtol=1e-4, atol=1e-6)
    c_loss = ref_loss.safe_ppo_value(out["c_values"], batch["c_returns"])
    np.testing.assert_allclose(c_loss.item(), g["c_value_loss"], rtol=1e-4)
    (total + c_loss).backward()
    named = dict(oracle_model.named_parameters())
    assert len(g["grad_names"]) == 252
    for n in g["grad_names"]:
        nrm, prj = grad_probe(str(n), named[str(n)].grad)
        want = g["gp:" + str(n)]
        np.testing.assert_allclose([nrm, prj], want, rtol=2e-3, atol=1e-5 + 1e-4 * want[0], err_msg=str(n))
    # exactly the parameters the reference gives gradients to (the frozen T5 gets none)
    have = {n for n, p in named.items() if p.grad is not None and float(p.grad.abs().sum()) > 0}
    assert have == {str(n) for n in g["grad_names"]}


def test_g5_acting_matches_update_path(oracle_model):
    g, gu = _load("g5_acting.npz"), _load("g5_samelen.npz")
    obs = {k[4:]: torch.from_numpy(v) for k, v in gu.items() if k.startswith("obs:")}
    pa, masks = torch.from_numpy(gu["prev_actions"]), torch.from_numpy(gu["masks"])
    for tw in (oracle_model, oracle_model.critic_tsfm, oracle_model.c_critic_tsfm):
        tw.time_step_counter = 0
    outs = []
    with torch.no_grad():
        for t in range(pa.shape[0]):
            o, _ = oracle_model({k: v[t : t + 1] for k, v in obs.items()}, None, pa[t : t + 1], masks[t : t + 1])
            outs.append(o)
    lg = torch.cat([ref_loss.categorical(o["logits"]) for o in outs])
    np.testing.assert_allclose(lg.numpy(), g["logits"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.cat([o["values"] for o in outs]).numpy(), g["values"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(torch.cat([o["c_values"] for o in outs]).numpy(), g["c_values"], rtol=1e-4, atol=2e-5)
    # reference property: with equal token lengths acting == update path
    np.testing.assert_allclose(g["logits"], gu["logits"], rtol=1e-3, atol=1e-4)


def test_g6_losses():
    g = _load("g6_losses.npz")
    batch = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("batch:")}
    for clipped in (False, True):
        for lam in (0.0, 0.37, 5.0):
            r = torch.from_numpy(g["raw_logits"]).requires_grad_(True)
            v = torch.from_numpy(g["values_pred"]).requires_grad_(True)
            total, info = ref_loss.safe_ppo_log_grad(r, v, batch, lam, entropy_coef=0.01, use_clipped_value_loss=clipped)
            total.backward()
            key = f"safe:{int(clipped)}:{lam}"
            np.testing.assert_allclose([info[k] for k in ("ppo_total", "value", "action", "entropy")], g[key + ":scalars"], rtol=1e-5)
            np.testing.assert_allclose(r.grad.numpy(), g[key + ":dlogits"], rtol=1e-4, atol=1e-8)
            np.testing.assert_allclose(v.grad.numpy(), g[key + ":dvalues"], rtol=1e-4, atol=1e-8)
    r = torch.from_numpy(g["raw_logits"]).requires_grad_(True)
    v = torch.from_numpy(g["values_pred"]).requires_grad_(True)
    total, info = ref_loss.ppo_log_grad(r, v, batch, entropy_coef=0.01)
    total.backward()
    np.testing.assert_allclose([info[k] for k in ("ppo_total", "value", "action", "entropy")], g["ppo:scalars"], rtol=1e-5)
    np.testing.assert_allclose(r.grad.numpy(), g["ppo:dlogits"], rtol=1e-4, atol=1e-8)
    # lambda = 0  =>  SafePPOLogGrad == PPOLogGrad
    np.testing.assert_allclose(g["safe:0:0.0:scalars"], g["ppo:scalars"], rtol=1e-6)


def test_g8_imitation_learning_model_vs_reference():
    """oracle.ref_il.RefEarlyFusion (EarlyFusionCnnTransformer small_3 / llama decoder on pre-encoded features) vs the reference's
    own logits, cross-entropy loss and gradient checksums (tests/golden/make_golden_il.py)."""
    from oracle.detfill import fill_state_dict, grad_probe
    from oracle.ref_il import RefEarlyFusion

    g = dict(np.load(os.path.join(G, "g8_il.npz"), allow_pickle=False))
    m = RefEarlyFusion().eval()
    want = {l.split("\t")[0]: l.rstrip("\n").split("\t")[1] for l in open(os.path.join(G, "state_dict_manifest_il.txt"))}
    have = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert have == want
    fill_state_dict(m, seed=7, share_t5=False)
    batch = {k: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k])
             for k in ("raw_navigation_camera", "raw_manipulation_camera", "time_ids", "an_object_is_in_hand", "actions", "last_actions", "padding_mask")}
    batch["goals"] = dict(input_ids=torch.from_numpy(g["goal_ids"]), attention_mask=torch.from_numpy(g["goal_mask"]))
    out = m(batch)
    out["loss"].backward()
    assert np.allclose(out["actions_logits"].detach().numpy(), g["logits"], rtol=2e-4, atol=2e-4)
    assert abs(out["loss"].item() - float(g["loss"])) < 1e-5
    n = 0
    for name, p in m.named_parameters():
        if "gp:" + name in g:
            got = np.array(grad_probe(name, p.grad))
            assert np.allclose(got, g["gp:" + name], rtol=2e-3, atol=1e-6), (name, got, g["gp:" + name])
            n += 1
    assert n == sum(k.startswith("gp:") for k in g)


def test_g9_imitation_learning_siglip_preset_vs_reference():
    """oracle.ref_il.RefEarlyFusion in the ``siglip_base_3`` configuration vs the reference's own model around the (third-party, restated)
    SigLIP text tower: [tokens, pooled] concatenation, 768 -> 512 text adapter, 768-channel compressor, fusion over 1 + 168 + 65 tokens."""
    from oracle.detfill import fill_state_dict, grad_probe
    from oracle.ref_il import RefEarlyFusion

    g = dict(np.load(os.path.join(G, "g9_il_siglip.npz"), allow_pickle=False))
    m = RefEarlyFusion(dino_dim=768, text_encoder="SigLIPBase").eval()
    want = {l.split("\t")[0]: l.rstrip("\n").split("\t")[1] for l in open(os.path.join(G, "state_dict_manifest_il_siglip.txt"))}
    have = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert have == want
    fill_state_dict(m, seed=11, share_t5=False)
    batch = {k: torch.from_numpy(g[k].astype(np.float32) if g[k].dtype == np.float16 else g[k])
             for k in ("raw_navigation_camera", "raw_manipulation_camera", "time_ids", "an_object_is_in_hand", "actions", "last_actions", "padding_mask")}
    batch["goals"] = torch.from_numpy(g["goal_ids"])
    out = m(batch)
    out["loss"].backward()
    assert np.allclose(out["actions_logits"].detach().numpy(), g["logits"], rtol=2e-4, atol=2e-4)
    assert abs(out["loss"].item() - float(g["loss"])) < 1e-5
    n = 0
    for name, p in m.named_parameters():
        if "gp:" + name in g:
            got = np.array(grad_probe(name, p.grad))
            assert np.allclose(got, g["gp:" + name], rtol=2e-3, atol=1e-6), (name, got, g["gp:" + name])
            n += 1
    assert n == sum(k.startswith("gp:") for k in g) == 84
