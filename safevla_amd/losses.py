"""Loss mirrors with the reference's constructor / ``loss(...)`` signatures, computed by the fused HIP kernels.

  * ``SafePPOLogGrad`` / ``PPOLogGrad``   /root/reference/training/online/loss/customized_loss.py:301-449 / :163-298
  * ``PPOValue`` / ``SafePPOValue``        [3P AllenAct fork] call sites /root/reference/training/online/dinov2_vits_tsfm_base.py:337-342
  * ``HLGaussLoss``                        /root/reference/utils/loss_functions.py:7-30 (critic_type == "discrete"; fused kernel)

``loss(step_count, batch, actor_critic_output, **kwargs) -> (total_loss, info)``: ``total_loss`` is a 0-d tensor attached
to autograd (one fused forward+backward kernel launch; the backward just scales the pre-computed gradients), ``info`` has
the reference's keys (``ppo_total, value, action, entropy, ...``).
"""
import math
from typing import Callable, Dict, Optional, Tuple

import torch

from . import ops


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, values, total, dlogits, dvalues):
        ctx.save_for_backward(dlogits, dvalues)
        return total.clone()

    @staticmethod
    def backward(ctx, g):
        dl, dv = ctx.saved_tensors
        return (dl * g if dl is not None else None), (dv * g if dv is not None else None), None, None, None


class SafePPOLogGrad:
    def __init__(self, clip_param: float, value_loss_coef: float, entropy_coef: float, use_clipped_value_loss: bool = True,
                 action_loss_schedule: Optional[Callable[[int], float]] = None, discrete_critics: bool = False,
                 normalize_advantage: bool = True, clip_decay: Optional[Callable[[int], float]] = None, **kw):
        self.discrete_critics = discrete_critics
        self.clip_param, self.value_loss_coef, self.entropy_coef = clip_param, value_loss_coef, entropy_coef
        self.use_clipped_value_loss = use_clipped_value_loss
        self.action_loss_schedule = action_loss_schedule if action_loss_schedule is not None else (lambda x: 1.0)
        self.clip_decay = clip_decay if clip_decay is not None else (lambda x: 1.0)
        self.adv_key = "norm_adv_targ" if normalize_advantage else "adv_targ"
        self.c_adv_key = "c_" + self.adv_key
        self.safe = True

    def loss(self, step_count: int, batch: Dict[str, torch.Tensor], actor_critic_output, *args, n_total: Optional[int] = None,
             **kwargs) -> Tuple[torch.Tensor, Dict[str, float]]:
        logits = actor_critic_output.distributions.raw_logits
        values = actor_critic_output.values
        T, B, A = logits.shape
        R = T * B
        lam = float(kwargs["lagrangian_multiplier"]) if self.safe else 0.0   # reference: .item() of a detached scalar
        clip = self.clip_param * self.clip_decay(step_count)
        aw = float(self.action_loss_schedule(step_count))
        inv_n = 1.0 / float(n_total if n_total is not None else R)
        f = lambda t: t.reshape(R).contiguous().float()
        for key in (self.adv_key,) + ((self.c_adv_key,) if self.safe else ()):
            if key not in batch:
                raise KeyError(f"{type(self).__name__}(normalize_advantage=True) reads batch['{key}']: build the batch with "
                               "RolloutStorage.batch_slice(..., normalized=True), or construct the loss with normalize_advantage=False "
                               "(what the reference config does: dinov2_vits_tsfm_base.py:314-322)")
        c_adv = f(batch[self.c_adv_key]) if self.safe else None
        ex = getattr(actor_critic_output, "extras", {}) or {}
        if self.discrete_critics:
            # customized_loss.py:364-370: value_loss = 0.5 * loss_func(extras["full_logits"], returns) -- NOT the ``values`` field.
            # (In the 3-tower wrapper ``extras`` are the cost-critic tower's, separate_actor_critic.py:31-36: the reference's own data flow.)
            full_logits, hl = ex["full_logits"], ex["loss_func"]
            NB = full_logits.shape[-1]
            sums, dl, _ = ops.ppo_lag_loss_fwd_bwd(
                logits.detach().reshape(R, A).contiguous(), f(values.detach()), batch["actions"].reshape(R).contiguous(),
                f(batch["old_action_log_probs"]), f(batch[self.adv_key]), c_adv, f(batch["returns"]), None, lam, clip, 0.0, aw,
                self.entropy_coef, False, inv_n)
            _, dfl, hs = ops.hlgauss_fwd_bwd(full_logits.detach().reshape(R, NB).contiguous().float(), f(batch["returns"]), None, hl.min_value,
                                             hl.max_value, hl.sigma, 0.5 * self.value_loss_coef, inv_n, want_values=False)
            s = sums * inv_n
            value, action, ent = 0.5 * hs[0] * inv_n, s[1], s[2]
            total = (self.value_loss_coef * value + aw * action + self.entropy_coef * ent).float()
            total = _FusedLoss.apply(logits, full_logits, total, dl.view(T, B, A), dfl.view_as(full_logits))
        else:
            sums, dl, dv = ops.ppo_lag_loss_fwd_bwd(
                logits.detach().reshape(R, A).contiguous(), f(values.detach()), batch["actions"].reshape(R).contiguous(),
                f(batch["old_action_log_probs"]), f(batch[self.adv_key]), c_adv, f(batch["returns"]),
                f(batch["values"]) if self.use_clipped_value_loss else None, lam, clip, self.value_loss_coef, aw, self.entropy_coef,
                self.use_clipped_value_loss, inv_n)
            s = sums * inv_n
            value, action, ent = 0.5 * s[0], s[1], s[2]
            total = (self.value_loss_coef * value + aw * action + self.entropy_coef * ent).float()
            total = _FusedLoss.apply(logits, values, total, dl.view(T, B, A), dv.view(T, B, 1))
        sc = torch.stack([total.detach().double(), value, action, ent]).cpu().tolist()   # one host sync (reference: four)
        info = {"ppo_total": sc[0], "value": sc[1], "action": sc[2], "entropy": sc[3], "action_weight": aw}
        info["weight_grad"] = torch.tensor([0.0])            # customized_loss.py:389-392: default when the model did not report it
        for k_src, k_dst in (("bias_norm", "bias_norm"), ("weight_norm", "weight_norm"), ("weight_grad_norm", "weight_grad")):
            if k_src in ex:
                info[k_dst] = ex[k_src]
        return total, info


class PPOLogGrad(SafePPOLogGrad):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.safe = False


class _ValueLoss:
    values_key, returns_key, info_key = "values", "returns", "value"

    def __init__(self, clip_param: float = 0.1, use_clipped_value_loss: bool = True, clip_decay: Optional[Callable[[int], float]] = None, **kw):
        """Upstream AllenAct ``PPOValue(clip_param, use_clipped_value_loss=True, clip_decay=None)`` [3P] -- same defaults; the reference
        instantiates it with ``use_clipped_value_loss=False`` through ``NewPPOConfig`` (dinov2_vits_tsfm_base.py:314-322,337-342), and so
        does every config in this repo (train.py, bench.py)."""
        self.clip_param, self.use_clipped_value_loss = clip_param, use_clipped_value_loss
        self.clip_decay = clip_decay if clip_decay is not None else (lambda x: 1.0)

    def loss(self, step_count, batch, actor_critic_output, *args, n_total: Optional[int] = None, **kwargs):
        v = getattr(actor_critic_output, self.values_key)
        R = v.numel()
        inv_n = 1.0 / float(n_total if n_total is not None else R)
        old = batch[self.values_key].reshape(R).contiguous().float() if self.use_clipped_value_loss else None
        sums, dv = ops.value_mse_fwd_bwd(v.detach().reshape(R).contiguous().float(), batch[self.returns_key].reshape(R).contiguous().float(), 1.0, inv_n,
                                         old_values=old, clip=self.clip_param * self.clip_decay(step_count))
        total = (0.5 * sums[0] * inv_n).float()
        total = _FusedLoss.apply(None, v, total, None, dv.view_as(v))
        return total, {self.info_key: float(total.detach())}


class PPOValue(_ValueLoss):
    pass


class SafePPOValue(_ValueLoss):
    values_key, returns_key, info_key = "c_values", "c_returns", "c_value"


class HLGaussLoss:
    """Histogram-Gaussian critic loss (utils/loss_functions.py:7-30), used when ``critic_type == "discrete"``
    (allenact_dino_transformer.py:152-159: min -5, max 15, 101 bins, sigma 0.15).  Device tensors run the fused HIP kernel
    (``svla_hlgauss_fwd_bwd_f32``: cross-entropy against the Gaussian-histogram target + its gradient in one launch); the closed forms
    below are the same arithmetic for host tensors (fixtures, configuration-time checks)."""

    def __init__(self, min_value: float, max_value: float, num_bins: int, sigma: float):
        self.min_value, self.max_value, self.num_bins, self.sigma = min_value, max_value, num_bins, sigma
        self.support = torch.linspace(min_value, max_value, num_bins + 1, dtype=torch.float32)

    def transform_to_probs(self, target):
        sup = self.support.to(target.device)
        cdf = torch.special.erf((sup - target.unsqueeze(-1)) / (math.sqrt(2.0) * self.sigma))
        return (cdf[..., 1:] - cdf[..., :-1]) / (cdf[..., -1] - cdf[..., 0]).unsqueeze(-1)

    def transform_from_probs(self, probs):
        sup = self.support.to(probs.device)
        return (probs * ((sup[:-1] + sup[1:]) / 2)).sum(-1)

    def value_from_logits(self, logits):
        """``transform_from_probs(softmax(logits))`` -- the read-out of DiscreteCriticHead.forward -- on the kernel."""
        flat = logits.detach().reshape(-1, logits.shape[-1]).contiguous().float()
        v, _, _ = ops.hlgauss_fwd_bwd(flat, None, None, self.min_value, self.max_value, self.sigma, want_grad=False)
        return v.view(logits.shape[:-1])

    def __call__(self, logits, target):
        if not logits.is_cuda:
            return -(self.transform_to_probs(target) * torch.log_softmax(logits, -1)).sum(-1).mean()
        R = target.numel()
        flat = logits.reshape(R, logits.shape[-1])
        _, dl, sums = ops.hlgauss_fwd_bwd(flat.detach().contiguous().float(), target.reshape(R).contiguous().float(), None, self.min_value,
                                          self.max_value, self.sigma, 1.0, 1.0 / R, want_values=False)
        total = (sums[0] / R).float()
        return _FusedLoss.apply(None, flat, total, None, dl)

    forward = __call__
