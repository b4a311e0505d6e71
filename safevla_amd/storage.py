"""Device-resident rollout storage with reward + cost GAE.

Mirrors the call contract of the AllenAct-fork ``RolloutBlockStorage`` [3P; not in /root/reference] as it is used by
the reference: ``initialize(observations, num_samplers, recurrent_memory_specification, action_space)``,
``add(observations, memory, actions, action_log_probs, value_preds, rewards, costs, c_value_preds, masks)``,
``agent_input_for_next_step()``, ``after_updates()``
(/root/reference/architecture/models/allenact_transformer_models/inference_agent.py:246-269,174,286), plus
``compute_returns`` (gamma / tau: /root/reference/training/online/dinov2_vits_tsfm_base.py:345-347) and
``batched_experience_generator`` (keys consumed by SafePPOLogGrad: training/online/loss/customized_loss.py:327-385).

HBM layout (MI355X-first): everything lives in preallocated [T+1, B, ...] / [T, B] device tensors; DINO features can
be held as bf16 tokens [T+1, B, 2, 84, 384] (``dino_tokens``, 2.7x smaller than fp32 channels-first and already in
the layout the compressor GEMM wants); GAE for rewards and costs is one fused kernel launch.
"""
from typing import Dict, Iterator, Optional

import torch

from . import ops


class RolloutStorage:
    def __init__(self, num_steps: int, device="cuda", store_tokens: bool = True,
                 nav_uuid="rgb_dinov2", manip_uuid="manipulation_rgb_dinov2"):
        self.T = num_steps
        self.device = torch.device(device)
        self.store_tokens = store_tokens
        self.nav_uuid, self.manip_uuid = nav_uuid, manip_uuid
        self.step = 0
        self.B = 0
        self.observations: Dict[str, torch.Tensor] = {}

    # ---- fork API ---------------------------------------------------------------------------------------------
    def initialize(self, observations: Dict[str, torch.Tensor], num_samplers: Optional[int] = None,
                   recurrent_memory_specification=None, action_space=None, **kw):
        """``observations``: first step, tensors shaped [B, ...] (or [1, B, ...])."""
        T, dev = self.T, self.device
        obs = {k: (v[0] if v.dim() > 1 and num_samplers is not None and v.shape[0] == 1 and v.shape[1] == num_samplers and v.dim() > 2 else v)
               for k, v in observations.items()}
        B = num_samplers if num_samplers is not None else next(iter(obs.values())).shape[0]
        self.B = B
        self.observations = {}
        for k, v in obs.items():
            if self.store_tokens and k in (self.nav_uuid, self.manip_uuid):
                continue
            self.observations[k] = torch.zeros((T + 1, B) + tuple(v.shape[1:]), device=dev, dtype=v.dtype)
        if self.store_tokens and self.nav_uuid in obs:
            self.observations["dino_tokens"] = torch.zeros(T + 1, B, 2, 84, 384, device=dev, dtype=torch.bfloat16)
        f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        self.masks = f(T + 1, B, 1)
        self.value_preds, self.c_value_preds = f(T + 1, B, 1), f(T + 1, B, 1)
        self.returns, self.c_returns = f(T + 1, B, 1), f(T + 1, B, 1)
        self.rewards, self.costs = f(T, B, 1), f(T, B, 1)
        self.action_log_probs = f(T, B)
        self.actions = torch.zeros(T, B, device=dev, dtype=torch.int64)
        self.prev_actions = torch.zeros(T + 1, B, device=dev, dtype=torch.int64)
        self.adv_targ = self.c_adv_targ = None
        self._norm = None
        self.step = 0
        self._insert_obs(obs, 0)
        self.masks[0] = 0.0   # first step of a fresh storage: no previous action (AllenAct convention)

    def _insert_obs(self, obs, t):
        for k, v in obs.items():
            if self.store_tokens and k in (self.nav_uuid, self.manip_uuid):
                cam = 0 if k == self.nav_uuid else 1
                ops.feat_to_tokens(v.reshape(self.B, 384, 84).contiguous().float(), self.observations["dino_tokens"][t], cam)
            else:
                self.observations[k][t].copy_(v.reshape(self.observations[k][t].shape))

    def add(self, observations, memory, actions, action_log_probs, value_preds, rewards, costs, c_value_preds, masks):
        t = self.step
        assert t < self.T, "rollout storage is full: call after_updates()"
        B = self.B
        self._insert_obs({k: (v[0] if v.dim() > 1 and v.shape[0] == 1 and v.shape[1] == B and v.dim() > 2 else v) for k, v in observations.items()}, t + 1)
        self.actions[t].copy_(actions.reshape(B))
        self.prev_actions[t + 1].copy_(actions.reshape(B))
        self.action_log_probs[t].copy_(action_log_probs.reshape(B))
        self.value_preds[t].copy_(value_preds.reshape(B, 1))
        self.c_value_preds[t].copy_(c_value_preds.reshape(B, 1))
        self.rewards[t].copy_(rewards.reshape(B, 1))
        self.costs[t].copy_(costs.reshape(B, 1))
        self.masks[t + 1].copy_(masks.reshape(B, 1))
        self.step = t + 1

    def agent_input_for_next_step(self):
        t = self.step
        return dict(observations={k: v[t:t + 1] for k, v in self.observations.items()}, memory=None,
                    prev_actions=self.prev_actions[t:t + 1], masks=self.masks[t:t + 1])

    def after_updates(self):
        for v in self.observations.values():
            v[0].copy_(v[self.step])
        self.masks[0].copy_(self.masks[self.step])
        self.prev_actions[0].copy_(self.prev_actions[self.step])
        self.step = 0

    # ---- returns / advantages -------------------------------------------------------------------------------
    def compute_returns(self, next_value: torch.Tensor, next_c_value: torch.Tensor, use_gae: bool = True, gamma: float = 0.99,
                        tau: float = 0.95):
        """Fused reward+cost GAE on the GPU; fills returns/c_returns[:-1] and adv_targ/c_adv_targ ([T,B,1])."""
        assert use_gae, "the shipped pipeline uses GAE (use_gae=True)"
        T, B = self.T, self.B
        ret, adv, c_ret, c_adv = ops.gae_scan(self.rewards.view(T, B), self.costs.view(T, B), self.value_preds[:T].view(T, B),
                                              self.c_value_preds[:T].view(T, B), self.masks.view(T + 1, B),
                                              next_value.reshape(B).contiguous().float(), next_c_value.reshape(B).contiguous().float(),
                                              gamma, tau)
        self.value_preds[T].copy_(next_value.reshape(B, 1))
        self.c_value_preds[T].copy_(next_c_value.reshape(B, 1))
        self.returns[:T].copy_(ret.view(T, B, 1))
        self.c_returns[:T].copy_(c_ret.view(T, B, 1))
        self.adv_targ, self.c_adv_targ = adv.view(T, B, 1), c_adv.view(T, B, 1)
        self._norm = None

    def normalized_advantages(self):
        """``norm_adv_targ`` = (adv - mean) / (std + 1e-5) with the moments taken over ALL samplers of all ranks (upstream AllenAct's
        ``adv_stats_callback`` [3P]; unbiased std like ``Tensor.std()``), same for the cost advantages.  Lazy: the shipped pipeline
        runs with ``normalize_advantage=False`` (dinov2_vits_tsfm_base.py:321) and never asks for it."""
        if self._norm is None:
            from . import parallel

            out = []
            for a in (self.adv_targ, self.c_adv_targ):
                ad = a.double()
                mom = torch.stack([ad.sum(), (ad * ad).sum(), torch.tensor(float(a.numel()), device=a.device, dtype=torch.float64)])
                parallel.allreduce_sum_(mom)
                mean = mom[0] / mom[2]
                var = (mom[1] - mom[2] * mean * mean) / torch.clamp(mom[2] - 1, min=1.0)
                out.append(((ad - mean) / (var.clamp(min=0).sqrt() + 1e-5)).float())
            self._norm = tuple(out)
        return self._norm

    def batched_experience_generator(self, num_mini_batch: int = 1, generator: Optional[torch.Generator] = None,
                                     normalized: bool = True) -> Iterator[Dict]:
        """Mini-batches split the *env* axis into contiguous groups (random order), all T steps of each group."""
        T, B = self.T, self.B
        assert B >= num_mini_batch
        order = torch.randperm(num_mini_batch, generator=generator).tolist() if num_mini_batch > 1 else [0]
        bounds = [round(i * B / num_mini_batch) for i in range(num_mini_batch + 1)]
        for i in order:
            yield self.batch_slice(bounds[i], bounds[i + 1], normalized=normalized)

    def batch_slice(self, b0: int, b1: int, normalized: bool = False) -> Dict:
        """``normalized``: also provide ``norm_adv_targ`` / ``c_norm_adv_targ`` (a loss built with ``normalize_advantage=True`` reads
        them; without them it raises a KeyError instead of silently training on raw advantages)."""
        T = self.T
        s = slice(b0, b1)
        extra = {}
        if normalized:
            na, nca = self.normalized_advantages()
            extra = dict(norm_adv_targ=na[:, s], c_norm_adv_targ=nca[:, s])
        return dict(**extra, 
            observations={k: v[:T, s] for k, v in self.observations.items()}, memory=None, prev_actions=self.prev_actions[:T, s],
            masks=self.masks[:T, s], actions=self.actions[:, s], old_action_log_probs=self.action_log_probs[:, s],
            values=self.value_preds[:T, s], c_values=self.c_value_preds[:T, s], returns=self.returns[:T, s],
            c_returns=self.c_returns[:T, s], adv_targ=self.adv_targ[:, s], c_adv_targ=self.c_adv_targ[:, s],
            bsize=T * (b1 - b0))
