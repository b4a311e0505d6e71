"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the engine pieces that live in un-vendored deps.

**PARITY UNPINNED.**  None of this arithmetic is under /root/reference: it lives in the AllenAct fork
``git+https://github.com/Ethyn13/allenact.git@main`` (README.md:85, unpinned commit) and in
``omnisafe==0.5.0`` (requirements.txt:131).  Restated from the published upstream algorithms and
anchored on the reference's call sites:
  * GAE / ``compute_returns``: gamma=0.99, tau=0.95, use_gae=True (training/online/dinov2_vits_tsfm_base.py:345-347);
    storage ``add(... rewards, costs, c_value_preds, masks)`` argument list (inference_agent.py:255-267).
  * ``Lagrange``: imported at training/online/loss/customized_loss.py:14; ``cost_limit`` plumbed at
    training/online/allenact_trainer.py:22,71; consumed as ``lagrangian_multiplier`` (customized_loss.py:428).
  * update loop: Adam(lr), max_grad_norm=0.5, update_repeats=4, num_mini_batch=1 (dinov2_vits_tsfm_base.py:331-334).
Property tests (tests/test_oracle_props.py): GAE == O(T^2) definition; lambda monotone in Jc - limit.
"""
import math
from typing import Dict

import torch


def gae_scan(rewards, values, masks, next_value, gamma=0.99, tau=0.95):
    """rewards (T,B,1); values (T,B,1) = V(s_t); masks (T+1,B,1) with masks[t+1] = 1 - done_t;
    next_value (B,1) = V(s_T).  Returns (returns (T,B,1), adv (T,B,1))."""
    T = rewards.shape[0]
    V = torch.cat([values, next_value.unsqueeze(0)], dim=0)
    ret = torch.zeros_like(rewards)
    g = torch.zeros_like(next_value)
    for t in reversed(range(T)):
        delta = rewards[t] + gamma * V[t + 1] * masks[t + 1] - V[t]
        g = delta + gamma * tau * masks[t + 1] * g
        ret[t] = g + V[t]
    return ret, ret - values


def gae_definition(rewards, values, masks, next_value, gamma=0.99, tau=0.95):
    """O(T^2) textbook form: A_t = sum_k (gamma*tau)^k * prod_{j<=k} m_{t+j} * delta_{t+k}."""
    T, B = rewards.shape[:2]
    V = torch.cat([values, next_value.unsqueeze(0)], dim=0).double()
    r, m = rewards.double(), masks.double()
    delta = r + gamma * V[1:] * m[1:] - V[:-1]
    adv = torch.zeros_like(delta)
    for t in range(T):
        w = torch.ones_like(delta[0])
        for k in range(t, T):
            if k > t:
                w = w * gamma * tau * m[k]
            adv[t] += w * delta[k]
    return (adv + V[:-1]).float(), adv.float()


class RefLagrange:
    """omnisafe.common.lagrange.Lagrange (0.5.0): lambda is a scalar parameter updated by its own optimiser on
    loss = -lambda * (Jc - cost_limit), then clamped to [0, upper].  Defaults follow omnisafe PPOLag:
    init 0.001, lr 0.035, Adam."""

    def __init__(self, cost_limit, init=0.001, lr=0.035, optimizer="Adam", upper=None):
        self.cost_limit = float(cost_limit)
        self.lam = torch.nn.Parameter(torch.tensor(max(float(init), 0.0)))
        self.opt = getattr(torch.optim, optimizer)([self.lam], lr=lr)
        self.upper = upper

    def update(self, Jc: float) -> float:
        self.opt.zero_grad()
        loss = -self.lam * (float(Jc) - self.cost_limit)
        loss.backward()
        self.opt.step()
        self.lam.data.clamp_(0.0, self.upper)
        return float(self.lam.item())


def ref_update_losses(out: Dict[str, torch.Tensor], batch, lam: float, stage_losses, cfg=None):
    """Sum of the stage's named losses with unit weights (engine restatement)."""
    from . import ref_loss

    cfg = cfg or {}
    total = 0.0
    info = {}
    for name in stage_losses:
        if name == "ppo_log_loss":
            l, i = ref_loss.safe_ppo_log_grad(out["logits"], out["values"], batch, lam, **cfg)
            info.update(i)
        elif name == "ppo_value_loss":
            l = ref_loss.ppo_value(out["values"], batch["returns"])
            info["ppo_value/value"] = l.item()
        elif name == "safe_ppo_value_loss":
            l = ref_loss.safe_ppo_value(out["c_values"], batch["c_returns"])
            info["safe_ppo_value/c_value"] = l.item()
        else:
            raise KeyError(name)
        total = total + l
    return total, info


def global_grad_norm(params) -> float:
    return math.sqrt(sum(float(p.grad.double().pow(2).sum()) for p in params if p.grad is not None))
