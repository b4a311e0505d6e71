"""Loss mirrors with the reference's constructor / ``loss(...)`` signatures, computed by the fused HIP kernels.

  * ``SafePPOLogGrad`` / ``PPOLogGrad``   /root/reference/training/online/loss/customized_loss.py:301-449 / :163-298
  * ``PPOValue`` / ``SafePPOValue``        [3P AllenAct fork] call sites /root/reference/training/online/dinov2_vits_tsfm_base.py:337-342
  * ``HLGaussLoss``                        /root/reference/utils/loss_functions.py:7-30 (optional discrete critic; host math)

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
        if discrete_critics:
            raise NotImplementedError("discrete (HL-Gauss) critics are off in the shipped pipeline (critic_type='linear')")
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
        c_adv = f(batch[self.c_adv_key]) if self.safe else None
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
        ex = getattr(actor_critic_output, "extras", {}) or {}
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

    def __init__(self, clip_param: float = 0.1, use_clipped_value_loss: bool = False, **kw):
        if use_clipped_value_loss:
            raise NotImplementedError("clipped value loss is off in the shipped pipeline")

    def loss(self, step_count, batch, actor_critic_output, *args, n_total: Optional[int] = None, **kwargs):
        v = getattr(actor_critic_output, self.values_key)
        R = v.numel()
        inv_n = 1.0 / float(n_total if n_total is not None else R)
        sums, dv = ops.value_mse_fwd_bwd(v.detach().reshape(R).contiguous().float(), batch[self.returns_key].reshape(R).contiguous().float(), 1.0, inv_n)
        total = (0.5 * sums[0] * inv_n).float()
        total = _FusedLoss.apply(None, v, total, None, dv.view_as(v))
        return total, {self.info_key: float(total.detach())}


class PPOValue(_ValueLoss):
    pass


class SafePPOValue(_ValueLoss):
    values_key, returns_key, info_key = "c_values", "c_returns", "c_value"


class HLGaussLoss:
    """Histogram-Gaussian critic loss (utils/loss_functions.py:7-30); only used when critic_type == 'discrete'."""

    def __init__(self, min_value: float, max_value: float, num_bins: int, sigma: float):
        self.sigma = sigma
        self.support = torch.linspace(min_value, max_value, num_bins + 1, dtype=torch.float32)

    def transform_to_probs(self, target):
        sup = self.support.to(target.device)
        cdf = torch.special.erf((sup - target.unsqueeze(-1)) / (math.sqrt(2.0) * self.sigma))
        return (cdf[..., 1:] - cdf[..., :-1]) / (cdf[..., -1] - cdf[..., 0]).unsqueeze(-1)

    def transform_from_probs(self, probs):
        sup = self.support.to(probs.device)
        return (probs * ((sup[:-1] + sup[1:]) / 2)).sum(-1)

    def __call__(self, logits, target):
        return -(self.transform_to_probs(target) * torch.log_softmax(logits, -1)).sum(-1).mean()
