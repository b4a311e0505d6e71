#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE ITSELF (imported from /root/reference) on CPU.

Runs only in the build container (the reference never travels); the emitted ``*.npz`` files are data:
seeded inputs + the reference's outputs.  Third-party packages the reference imports but the image lacks
(allenact, gym, omnisafe, open_clip, ...) are replaced by ``sys.modules`` shims that restate only the
tiny pieces touched at import/forward time (heads, output containers, PPO.__init__).  Weights on both
sides come from ``oracle.detfill.fill_state_dict`` (name-seeded), so fixtures hold no parameters.

    python tests/golden/make_golden.py            # writes tests/golden/g*.npz
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.detfill import fill_state_dict, grad_probe  # noqa: E402
from safevla_amd.text import GoalTokenizer, str_to_bytes  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path=None):
    m = _mod(name)
    m.__path__ = [path] if path else []
    return m


def install_shims():
    # ---- gym ---------------------------------------------------------------------------------
    class Discrete:
        def __init__(self, n):
            self.n = n

    class Box:
        def __init__(self, low=0, high=1, shape=(), dtype=np.float32):
            self.shape = tuple(shape)

    class SpaceDict:
        def __init__(self, spaces):
            self.spaces = dict(spaces)

    gym = _pkg("gym")
    gym.spaces = _mod("gym.spaces", Discrete=Discrete, Box=Box, Dict=SpaceDict)

    # ---- allenact (restating only what is touched) ------------------------------------------------
    class CategoricalDistr(torch.distributions.Categorical):
        def log_prob(self, value):
            if value.shape == self.logits.shape[:-1]:
                return self.logits.gather(-1, value.unsqueeze(-1)).squeeze(-1)
            return self.logits.gather(-1, value).squeeze(-1)

    class LinearActorHead(nn.Module):
        def __init__(self, num_inputs, num_outputs):
            super().__init__()
            self.linear = nn.Linear(num_inputs, num_outputs)
            nn.init.orthogonal_(self.linear.weight, gain=0.01)
            nn.init.constant_(self.linear.bias, 0)

        def forward(self, x):
            return CategoricalDistr(logits=self.linear(x))

    class LinearCriticHead(nn.Module):
        def __init__(self, input_size):
            super().__init__()
            self.fc = nn.Linear(input_size, 1)
            nn.init.orthogonal_(self.fc.weight)
            nn.init.constant_(self.fc.bias, 0)

        def forward(self, x):
            return self.fc(x).view(*x.shape[:2], -1)

    class ActorCriticOutput:
        def __init__(self, distributions, values, extras):
            self.distributions, self.values, self.extras = distributions, values, extras

        def __class_getitem__(cls, item):
            return cls

    class SafeActorCriticOutput(ActorCriticOutput):
        def __init__(self, distributions, values, c_values, extras):
            super().__init__(distributions, values, extras)
            self.c_values = c_values

    class VisualNavActorCritic(nn.Module):
        def __init__(self, action_space, observation_space, hidden_size=512, multiple_beliefs=False,
                     beliefs_fusion=None, auxiliary_uuids=None, **kw):
            super().__init__()
            self.action_space, self.observation_space = action_space, observation_space
            self._hidden_size = hidden_size
            self.multiple_beliefs, self.beliefs_fusion = multiple_beliefs, beliefs_fusion
            self.auxiliary_uuids = auxiliary_uuids
            self.aux_models = nn.ModuleDict()

        def create_aux_models(self, obs_embed_size, action_embed_size):
            pass

    class AbstractActorCriticLoss:
        def __init__(self, *a, **k):
            pass

    class PPO(AbstractActorCriticLoss):
        def __init__(self, clip_param, value_loss_coef, entropy_coef, use_clipped_value_loss=True,
                     clip_decay=None, entropy_method_name="entropy", normalize_advantage=True,
                     show_ratios=False, *a, **k):
            super().__init__()
            self.clip_param, self.value_loss_coef, self.entropy_coef = clip_param, value_loss_coef, entropy_coef
            self.use_clipped_value_loss = use_clipped_value_loss
            self.clip_decay = clip_decay if clip_decay is not None else (lambda x: 1.0)
            self.entropy_method_name = entropy_method_name
            self.show_ratios = show_ratios
            self.adv_key = "norm_adv_targ" if normalize_advantage else "adv_targ"

    import logging

    _pkg("allenact")
    _pkg("allenact.algorithms")
    _pkg("allenact.algorithms.onpolicy_sync")
    _mod("allenact.algorithms.onpolicy_sync.policy", LinearActorHead=LinearActorHead, LinearCriticHead=LinearCriticHead,
         DistributionType=object, ObservationType=dict)
    lp = _pkg("allenact.algorithms.onpolicy_sync.losses")
    lp.PPO = PPO
    _mod("allenact.algorithms.onpolicy_sync.losses.abstract_loss", AbstractActorCriticLoss=AbstractActorCriticLoss,
         ObservationType=dict)
    _pkg("allenact.base_abstractions")
    _mod("allenact.base_abstractions.misc", ActorCriticOutput=ActorCriticOutput, SafeActorCriticOutput=SafeActorCriticOutput,
         Memory=dict)
    _mod("allenact.base_abstractions.distributions", Distr=object, CategoricalDistr=CategoricalDistr)
    _pkg("allenact.embodiedai")
    _pkg("allenact.embodiedai.aux_losses")
    _mod("allenact.embodiedai.aux_losses.losses", MultiAuxTaskNegEntropyLoss=type("M", (), {"UUID": "multitask_entropy"}))
    _pkg("allenact.embodiedai.models")
    _mod("allenact.embodiedai.models.visual_nav_models", VisualNavActorCritic=VisualNavActorCritic, FusionType=object)
    _pkg("allenact.utils")
    _mod("allenact.utils.system", get_logger=lambda: logging.getLogger("ref"))
    _pkg("omnisafe")
    _pkg("omnisafe.common")
    _mod("omnisafe.common.lagrange", Lagrange=object)
    oc = _pkg("open_clip")
    oc.create_model_from_pretrained = None
    _mod("open_clip.transformer", TextTransformer=object)
    _pkg("clip")

    # ---- reference packages: real files, but skip heavyweight package __init__s --------------------
    for p in ("architecture", "architecture/models", "architecture/models/transformer_models",
              "architecture/models/allenact_transformer_models", "training", "training/online",
              "training/online/third_party_models", "training/online/third_party_models/llama",
              "training/online/loss", "utils"):
        _pkg(p.replace("/", "."), os.path.join(REF, p))
    _mod("architecture.models.transformer_models.image_encoders", IMAGE_ENCODERS={})
    _mod("utils.sensor_constant_utils", is_a_visual_sensor=lambda s: True)
    _mod("utils.bbox_utils", get_best_of_two_bboxes=None)
    _mod("utils.nn_utils", debug_model_info=lambda *a, **k: None)
    from safevla_amd.text import bytes_to_str

    _mod("utils.string_utils", convert_byte_to_string=lambda b, max_len=None: bytes_to_str(b))
    sys.path.insert(0, REF)
    return gym, SpaceDict, Box, Discrete


def build_reference_model():
    gym, SpaceDict, Box, Discrete = install_shims()
    from transformers import T5Config, T5EncoderModel

    class _T5:
        @staticmethod
        def from_pretrained(name):
            cfg = T5Config(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8,
                           feed_forward_proj="relu")
            return T5EncoderModel(cfg)

    class _Tok:
        @staticmethod
        def from_pretrained(name):
            return GoalTokenizer()

    adt = importlib.import_module("architecture.models.allenact_transformer_models.allenact_dino_transformer")
    adt.T5EncoderModel, adt.AutoTokenizer = _T5, _Tok
    sep = importlib.import_module("architecture.models.allenact_transformer_models.separate_actor_critic")
    obs_space = SpaceDict({
        "rgb_dinov2": Box(shape=(7, 12, 384)), "manipulation_rgb_dinov2": Box(shape=(7, 12, 384)),
        "natural_language_spec": Box(shape=(1000,)), "time_step": Box(), "traj_index": Box(),
        "an_object_is_in_hand": Box(shape=(1,)),
    })
    torch.manual_seed(0)
    model = sep.SafeDinoLLAMATxNavActorCriticSeparate(
        action_space=Discrete(20), observation_space=obs_space, goal_sensor_uuid="natural_language_spec",
        rgb_dino_preprocessor_uuid="rgb_dinov2", manipulation_rgb_dino_preprocessor_uuid="manipulation_rgb_dinov2",
        an_object_is_in_hand_uuid="an_object_is_in_hand", num_tx_layers=3, num_tx_heads=8, hidden_size=512,
        goal_dims=512, add_prev_actions=True, add_prev_action_null_token=True, auxiliary_uuids=[], max_steps=500,
        time_step_uuid="time_step", initial_tgt_cache_shape=(500, 4, 512), traj_idx_uuid="traj_index",
        traj_max_idx=2048, relevant_object_box_uuid=None, accurate_object_box_uuid=None, prev_checkpoint=None,
    )
    return model


# ------------------------------------------------------------------------------------------------
def synth_obs(T, B, goals, seed, done_p=0.12):
    """Seeded observation block with mid-rollout episode boundaries (shapes: SURVEY.md section 8 a5)."""
    rs = np.random.RandomState(seed)
    obs = {
        "rgb_dinov2": rs.standard_normal((T, B, 384, 7, 12)).astype(np.float32),
        "manipulation_rgb_dinov2": rs.standard_normal((T, B, 384, 7, 12)).astype(np.float32),
    }
    time_step = np.zeros((T, B), np.int64)
    traj = np.zeros((T, B), np.int64)
    masks = np.ones((T, B, 1), np.float32)
    goal = np.zeros((T, B, 1000), np.uint8)
    cur_t = rs.randint(0, 40, size=B)
    cur_traj = rs.randint(0, 2000, size=B)
    cur_goal = rs.randint(0, len(goals), size=B)
    for t in range(T):
        for b in range(B):
            if t > 0 and rs.rand() < done_p:
                cur_t[b] = 0
                cur_traj[b] = (cur_traj[b] + 1) % 2048
                cur_goal[b] = rs.randint(0, len(goals))
                masks[t, b, 0] = 0.0
            time_step[t, b] = cur_t[b]
            traj[t, b] = cur_traj[b]
            goal[t, b] = str_to_bytes(goals[cur_goal[b]])[: 1000]
            cur_t[b] += 1
    obs["time_step"], obs["traj_index"], obs["natural_language_spec"] = time_step, traj, goal
    obs["an_object_is_in_hand"] = rs.randint(0, 2, size=(T, B, 1)).astype(np.int64)
    prev_actions = rs.randint(0, 20, size=(T, B)).astype(np.int64)
    return obs, prev_actions, masks


def to_t(d):
    return {k: torch.from_numpy(v) for k, v in d.items()}


def synth_batch(T, B, seed):
    rs = np.random.RandomState(seed)
    return {
        "actions": rs.randint(0, 20, size=(T, B)).astype(np.int64),
        "old_action_log_probs": (-3.0 + 0.3 * rs.standard_normal((T, B))).astype(np.float32),
        "adv_targ": rs.standard_normal((T, B, 1)).astype(np.float32),
        "c_adv_targ": (0.5 * rs.standard_normal((T, B, 1))).astype(np.float32),
        "returns": (2.0 * rs.standard_normal((T, B, 1))).astype(np.float32),
        "values": (2.0 * rs.standard_normal((T, B, 1))).astype(np.float32),
    }


def main():
    out = {}
    torch.set_num_threads(8)
    model = build_reference_model()
    model.eval()
    fill_state_dict(model, seed=7)

    # ---- G1 HLGaussLoss ----------------------------------------------------------------------
    lf = importlib.import_module("utils.loss_functions")
    hl = lf.HLGaussLoss(min_value=-5.0, max_value=15.0, num_bins=101, sigma=0.15)
    rs = np.random.RandomState(1)
    tgt = torch.from_numpy(rs.uniform(-4.0, 14.0, 16).astype(np.float32))
    lg = torch.from_numpy(rs.standard_normal((16, 101)).astype(np.float32))
    pr = hl.transform_to_probs(tgt)
    np.savez_compressed(os.path.join(HERE, "g1_hlgauss.npz"), target=tgt.numpy(), logits=lg.numpy(), probs=pr.numpy(),
                        loss=hl(lg, tgt).numpy(), value=hl.transform_from_probs(torch.softmax(lg, -1)).numpy())

    # ---- G2 PositionalEncoder ------------------------------------------------------------------
    pos = torch.tensor([[0, 1, 2, 5, 17, 128, 499]])
    np.savez_compressed(os.path.join(HERE, "g2_posenc.npz"), pos=pos.numpy(), pe=model.time_encoder(pos).numpy())

    # ---- G3 llama decoder: full sequence (+grads) and KV-cache stepping ------------------------------
    B, T = 4, 32
    rs = np.random.RandomState(3)
    x = torch.from_numpy(rs.standard_normal((B, T, 512)).astype(np.float32)).requires_grad_(True)
    traj = torch.from_numpy(np.sort(rs.randint(0, 3, size=(B, T)), axis=1))
    mask = torch.tril(traj[:, :, None] == traj[:, None, :])[:, None]
    dec = model.decoder
    y = dec(x, 0, mask)
    w = torch.from_numpy(rs.standard_normal(tuple(y.shape)).astype(np.float32))
    for p in dec.parameters():
        p.grad = None
    (y * w).sum().backward()
    g3 = dict(x=x.detach().numpy(), traj=traj.numpy(), y=y.detach().numpy(), w=w.numpy(), dx=x.grad.numpy())
    for n, p in dec.named_parameters():
        g3["gp:" + n] = np.array(grad_probe(n, p.grad), np.float64)
    # KV-cache path: all envs start an episode at step 0 (time_step == counter) => full causal window
    with torch.no_grad():
        ys = [dec(x[:, t : t + 1].detach(), t, torch.ones(B, 1, 1, t + 1, dtype=torch.bool)) for t in range(T)]
        full = dec(x.detach(), 0, torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None].expand(B, 1, T, T))
    g3["y_cache"] = torch.cat(ys, dim=1).numpy()
    g3["y_causal"] = full.numpy()
    np.savez_compressed(os.path.join(HERE, "g3_decoder.npz"), **g3)

    # ---- G5 three-tower forward + SafePPOLogGrad + value losses, with gradient probes ---------------
    goals = ["find a mug", "navigate to the red apple", "pick up a bowl", "fetch the small blue vase now"]
    for tag, T, B, gl in (("g5_mixedlen", 8, 4, goals), ("g5_samelen", 6, 3, ["find a mug", "pick up bowl", "go to sofa"])):
        obs, pa, masks = synth_obs(T, B, gl, seed=11)
        batch = synth_batch(T, B, seed=12)
        batch["c_returns"] = (1.5 * np.random.RandomState(13).standard_normal((T, B, 1))).astype(np.float32)
        cl = importlib.import_module("training.online.loss.customized_loss")
        loss_fn = cl.SafePPOLogGrad(clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.0, use_clipped_value_loss=False,
                                    action_loss_schedule=None, discrete_critics=False, normalize_advantage=False)
        for p in model.parameters():
            p.grad = None
        aco, _ = model(to_t(obs), None, torch.from_numpy(pa), torch.from_numpy(masks))
        lam = 0.37
        total, info = loss_fn.loss(0, to_t(batch), aco, lagrangian_multiplier=torch.tensor(lam))
        c_loss = 0.5 * (torch.from_numpy(batch["c_returns"]) - aco.c_values).pow(2).mean()  # SafePPOValue restatement
        (total + c_loss).backward()
        g5 = {("obs:" + k): v for k, v in obs.items()}
        g5.update({("batch:" + k): v for k, v in batch.items()})
        g5.update(prev_actions=pa, masks=masks, lam=np.float32(lam), logits=aco.distributions.logits.detach().numpy(),
                  raw_logits=None, values=aco.values.detach().numpy(), c_values=aco.c_values.detach().numpy(),
                  ppo_total=np.float32(info["ppo_total"]), value=np.float32(info["value"]), action=np.float32(info["action"]),
                  entropy=np.float32(info["entropy"]), c_value_loss=c_loss.detach().numpy())
        g5.pop("raw_logits")
        names = []
        for n, p in model.named_parameters():
            if p.grad is not None and float(p.grad.abs().sum()) > 0:
                g5["gp:" + n] = np.array(grad_probe(n, p.grad), np.float64)
                names.append(n)
        g5["grad_names"] = np.array(names)
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **g5)
        print(tag, info, float(c_loss), len(names))

    # acting path (nsteps=1 with KV cache) vs update path on one rollout, same-length goals
    obs, pa, masks = synth_obs(6, 3, ["find a mug", "pick up bowl", "go to sofa"], seed=11)
    with torch.no_grad():
        outs = []
        # a T>1 call resets the per-tower counters; emulate a fresh rollout: envs whose episode is already
        # in progress at t=0 simply see an empty cache before it (same as the update-path mask).
        for tw in (model, model.critic_tsfm, model.c_critic_tsfm):
            tw.time_step_counter = 0
        for t in range(6):
            o = {k: torch.from_numpy(v[t : t + 1]) for k, v in obs.items()}
            aco, _ = model(o, None, torch.from_numpy(pa[t : t + 1]), torch.from_numpy(masks[t : t + 1]))
            outs.append((aco.distributions.logits.numpy(), aco.values.numpy(), aco.c_values.numpy()))
    np.savez_compressed(os.path.join(HERE, "g5_acting.npz"), logits=np.concatenate([o[0] for o in outs]),
                        values=np.concatenate([o[1] for o in outs]), c_values=np.concatenate([o[2] for o in outs]))

    # ---- G6 / G7 losses on (32,4) -----------------------------------------------------------------
    T, B = 32, 4
    rs = np.random.RandomState(21)
    batch = synth_batch(T, B, seed=22)
    raw = torch.from_numpy((1.5 * rs.standard_normal((T, B, 20))).astype(np.float32))
    vals = torch.from_numpy((2.0 * rs.standard_normal((T, B, 1))).astype(np.float32))
    g6 = {("batch:" + k): v for k, v in batch.items()}
    g6.update(raw_logits=raw.numpy(), values_pred=vals.numpy())
    Dist = sys.modules["allenact.base_abstractions.distributions"].CategoricalDistr
    ACO = sys.modules["allenact.base_abstractions.misc"].ActorCriticOutput
    extras = {"bias_norm": torch.zeros(1), "weight_norm": torch.zeros(1)}
    for clipped in (False, True):
        for lam in (0.0, 0.37, 5.0):
            lf_ = cl.SafePPOLogGrad(clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.01, use_clipped_value_loss=clipped,
                                    action_loss_schedule=None, discrete_critics=False, normalize_advantage=False)
            r, v = raw.clone().requires_grad_(True), vals.clone().requires_grad_(True)
            total, info = lf_.loss(0, to_t(batch), ACO(Dist(logits=r), v, dict(extras)), lagrangian_multiplier=torch.tensor(lam))
            total.backward()
            key = f"safe:{int(clipped)}:{lam}"
            g6[key + ":scalars"] = np.array([info["ppo_total"], info["value"], info["action"], info["entropy"]], np.float32)
            g6[key + ":dlogits"], g6[key + ":dvalues"] = r.grad.numpy(), v.grad.numpy()
    lf_ = cl.PPOLogGrad(clip_param=0.1, value_loss_coef=0.5, entropy_coef=0.01, use_clipped_value_loss=False,
                        action_loss_schedule=None, discrete_critics=False, normalize_advantage=False)
    r, v = raw.clone().requires_grad_(True), vals.clone().requires_grad_(True)
    total, info = lf_.loss(0, to_t(batch), ACO(Dist(logits=r), v, dict(extras)))
    total.backward()
    g6["ppo:scalars"] = np.array([info["ppo_total"], info["value"], info["action"], info["entropy"]], np.float32)
    g6["ppo:dlogits"], g6["ppo:dvalues"] = r.grad.numpy(), v.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g6_losses.npz"), **g6)

    # ---- state_dict manifest ------------------------------------------------------------------------
    with open(os.path.join(HERE, "state_dict_manifest.txt"), "w") as f:
        for k, v in model.state_dict().items():
            f.write(f"{k}\t{tuple(v.shape)}\n")
    print("done")


if __name__ == "__main__":
    main()
