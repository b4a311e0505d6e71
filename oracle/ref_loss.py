"""TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the PPO-Lagrangian losses.

  * ``safe_ppo_log_grad``  training/online/loss/customized_loss.py:317-449 (``SafePPOLogGrad``)   PINNED (G6)
  * ``ppo_log_grad``       training/online/loss/customized_loss.py:178-298 (``PPOLogGrad``)       PINNED (G7)
  * ``hl_gauss_*``         utils/loss_functions.py:7-30 (``HLGaussLoss``)                         PINNED (G1)
  * ``ppo_value`` / ``safe_ppo_value``  [3P AllenAct fork ``PPOValue``/``SafePPOValue``; call sites
    training/online/dinov2_vits_tsfm_base.py:337-342] -- **parity unpinned**, restated from upstream
    AllenAct: 0.5 * mean((returns - values)^2) with use_clipped_value_loss=False.
``CategoricalDistr`` [3P]: log-softmax-normalised logits, log_prob = gather, entropy = -sum p log p.
"""
import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


def categorical(logits: torch.Tensor):
    return logits - logits.logsumexp(dim=-1, keepdim=True)


def log_prob(logits, actions):
    return categorical(logits).gather(-1, actions.unsqueeze(-1)).squeeze(-1)


def entropy(logits):
    lp = categorical(logits)
    return -(lp.exp() * lp).sum(-1)


def safe_ppo_log_grad(
    logits: torch.Tensor,  # (T,B,A)
    values: torch.Tensor,  # (T,B,1)
    batch: Dict[str, torch.Tensor],
    lagrangian_multiplier: float,
    clip_param: float = 0.1,
    value_loss_coef: float = 0.5,
    entropy_coef: float = 0.0,
    use_clipped_value_loss: bool = False,
    action_weight: float = 1.0,
) -> Tuple[torch.Tensor, Dict[str, float]]:
    lp = log_prob(logits, batch["actions"])  # (T,B)
    neg_ent = -entropy(logits)
    ratio = torch.exp(lp - batch["old_action_log_probs"]).unsqueeze(-1)
    clamped = torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param)
    lam = float(lagrangian_multiplier)
    if lam == 0.0 and "c_adv_targ" not in batch:
        adv = batch["adv_targ"]
    else:
        adv = (batch["adv_targ"] - lam * batch["c_adv_targ"]) / (1.0 + lam)
    surr1, surr2 = ratio * adv, clamped * adv
    action_loss = -torch.where(surr2 < surr1, surr2, surr1)
    if use_clipped_value_loss:
        vc = batch["values"] + (values - batch["values"]).clamp(-clip_param, clip_param)
        value_loss = 0.5 * torch.max((values - batch["returns"]).pow(2), (vc - batch["returns"]).pow(2)).mean()
    else:
        value_loss = 0.5 * (batch["returns"] - values).pow(2).mean()
    a, e = action_loss.mean(), neg_ent.mean()
    total = value_loss * value_loss_coef + a * action_weight + e * entropy_coef
    return total, {"ppo_total": total.item(), "value": value_loss.item(), "action": a.item(), "entropy": e.item()}


def ppo_log_grad(logits, values, batch, **kw):
    b = dict(batch)
    b.pop("c_adv_targ", None)
    return safe_ppo_log_grad(logits, values, b, 0.0, **kw)


def ppo_value(values: torch.Tensor, returns: torch.Tensor) -> torch.Tensor:
    return 0.5 * (returns - values).pow(2).mean()


safe_ppo_value = ppo_value  # on (c_values, c_returns)


# ---- HLGaussLoss ------------------------------------------------------------------------------
def hl_support(min_value=-5.0, max_value=15.0, num_bins=101):
    return torch.linspace(min_value, max_value, num_bins + 1, dtype=torch.float32)


def hl_gauss_probs(target: torch.Tensor, support: torch.Tensor, sigma: float = 0.15):
    cdf = torch.special.erf((support - target.unsqueeze(-1)) / (math.sqrt(2.0) * sigma))
    z = cdf[..., -1] - cdf[..., 0]
    return (cdf[..., 1:] - cdf[..., :-1]) / z.unsqueeze(-1)


def hl_gauss_loss(logits, target, support, sigma=0.15):
    p = hl_gauss_probs(target, support, sigma)
    return -(p * F.log_softmax(logits, dim=-1)).sum(-1).mean()


def hl_gauss_value(probs, support):
    return (probs * ((support[:-1] + support[1:]) / 2)).sum(-1)
