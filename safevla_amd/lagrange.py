"""Lagrange multiplier of PPO-Lagrangian.

Mirrors ``omnisafe.common.lagrange.Lagrange`` (omnisafe==0.5.0, /root/reference/requirements.txt:131; imported at
/root/reference/training/online/loss/customized_loss.py:14; ``cost_limit`` plumbed at
/root/reference/training/online/allenact_trainer.py:22,71): lambda is a scalar optimised on
``loss = -lambda * (Jc - cost_limit)`` by its own optimiser and clamped to [0, upper].  The value is tiny scalar state,
kept on the host in fp32-faithful arithmetic and replicated deterministically on every rank (Jc is all-reduced first),
so no broadcast is needed.  PARITY UNPINNED for the fork's hyper-parameters (not in the reference tree): defaults are
omnisafe's PPOLag ones; everything is a constructor argument.
"""
import math

import numpy as np


class Lagrange:
    def __init__(self, cost_limit: float, lagrangian_multiplier_init: float = 0.001, lambda_lr: float = 0.035,
                 lambda_optimizer: str = "Adam", lagrangian_upper_bound=None):
        if lambda_optimizer not in ("Adam", "SGD"):
            raise ValueError(f"unsupported lambda_optimizer {lambda_optimizer}")
        self.cost_limit = float(cost_limit)
        self.lambda_lr = float(lambda_lr)
        self.lagrangian_upper_bound = lagrangian_upper_bound
        self.lambda_optimizer = lambda_optimizer
        self._lam = np.float32(max(float(lagrangian_multiplier_init), 0.0))
        self._m = np.float32(0.0)
        self._v = np.float32(0.0)
        self._t = 0

    @property
    def lagrangian_multiplier(self) -> float:
        return float(self._lam)

    def compute_lambda_loss(self, mean_ep_cost: float) -> float:
        return -float(self._lam) * (float(mean_ep_cost) - self.cost_limit)

    def update_lagrange_multiplier(self, Jc: float) -> float:
        f = np.float32
        g = f(-(float(Jc) - self.cost_limit))         # d loss / d lambda (difference in double, like the python scalars in omnisafe)
        if self.lambda_optimizer == "SGD":
            self._lam = f(self._lam - f(self.lambda_lr) * g)
        else:  # torch.optim.Adam defaults, fp32 like a torch scalar parameter
            b1, b2, eps = f(0.9), f(0.999), f(1e-8)
            self._t += 1
            self._m = f(self._m + (g - self._m) * f(1.0 - 0.9))          # lerp_(grad, 1 - beta1): weight is a double scalar
            self._v = f(f(self._v * b2) + f(f(1.0 - 0.999) * f(g * g)))   # mul_(beta2).addcmul_(g, g, value=1 - beta2)
            step_size = f(self.lambda_lr / (1.0 - 0.9 ** self._t))      # scalars in double, cast once (as torch does)
            bc2s = f(math.sqrt(1.0 - 0.999 ** self._t))
            denom = f(f(np.sqrt(self._v)) / bc2s + eps)
            self._lam = f(self._lam - step_size * f(self._m / denom))
        hi = np.inf if self.lagrangian_upper_bound is None else float(self.lagrangian_upper_bound)
        self._lam = f(min(max(float(self._lam), 0.0), hi))
        return float(self._lam)

    def state_dict(self):
        return dict(lam=float(self._lam), m=float(self._m), v=float(self._v), t=self._t)

    def load_state_dict(self, sd):
        self._lam, self._m, self._v, self._t = np.float32(sd["lam"]), np.float32(sd["m"]), np.float32(sd["v"]), int(sd["t"])
