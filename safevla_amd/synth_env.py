"""Synthetic rollout generator: replaces AI2-THOR with tensors of identical shapes / value ranges so the simulator is
off the critical path (BASELINE.json north_star).

Sensor contracts reproduced (paths relative to /root/reference):
  * two DINOv2 feature maps (384,7,12) per step    architecture/allenact_preprocessors/dino_preprocessors.py:31-35
  * ``time_step`` / ``traj_index`` (mod 2048)        environment/navigation_sensors.py:985-1042
  * ``an_object_is_in_hand`` int64[1]                environment/manipulation_sensors.py:10-26
  * goal: 1000-byte instruction or token ids         environment/navigation_sensors.py:144-183
  * 20 discrete actions, episode end at ``done`` or max_steps=500   training/online/base.py:129
  * reward +10 on success else 0; cost = #triggered safety predicates in {0..5}  tasks/object_nav_task.py:142-159,
    tasks/abstract_task.py:321-333
"""
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .storage import RolloutStorage

TASK_DONE_P = {"ObjectNav": 1.0 / 60, "PickUp": 1.0 / 42, "Fetch": 1.0 / 110}


@dataclass
class SynthSpec:
    T: int = 256
    B: int = 32
    L: int = 12                 # goal tokens (constant per episode)
    task: str = "Fetch"
    seed: int = 1234
    max_steps: int = 500
    cost_p: float = 0.05        # cost ~ Binomial(5, cost_p)


def fill_synthetic_rollout(model, spec: SynthSpec, device="cuda") -> Tuple[RolloutStorage, Dict[str, torch.Tensor], Dict[str, float]]:
    """Returns (storage filled with T steps, {next_value, next_c_value}, {episode_cost_sum, n_episodes})."""
    T, B, L = spec.T, spec.B, spec.L
    rs = np.random.RandomState(spec.seed)
    done_p = TASK_DONE_P.get(spec.task, 1.0 / 60)
    time_step = np.zeros((T + 1, B), np.int64)
    traj = np.zeros((T + 1, B), np.int64)
    masks = np.ones((T + 1, B, 1), np.float32)
    goal = np.zeros((T + 1, B, L), np.int64)
    rewards = np.zeros((T, B, 1), np.float32)
    costs = rs.binomial(5, spec.cost_p, size=(T, B, 1)).astype(np.float32)
    cur_t = rs.randint(0, 50, size=B)
    cur_traj = rs.randint(0, 2048, size=B)
    cur_goal = rs.randint(3, 32000, size=(B, L))
    cur_goal[:, -1] = 1  # EOS
    ep_cost = np.zeros(B)
    ep_cost_sum, n_ep = 0.0, 0
    masks[0] = 0.0
    for t in range(T + 1):
        time_step[t], traj[t], goal[t] = cur_t, cur_traj, cur_goal
        if t == T:
            break
        ep_cost += costs[t, :, 0]
        done = (rs.rand(B) < done_p) | (cur_t + 1 >= spec.max_steps)
        rewards[t, :, 0] = 10.0 * done * (rs.rand(B) < 0.5)
        masks[t + 1, :, 0] = 1.0 - done
        for b in np.nonzero(done)[0]:
            ep_cost_sum += ep_cost[b]; n_ep += 1; ep_cost[b] = 0.0
            cur_t[b] = -1
            cur_traj[b] = (cur_traj[b] + 1) % 2048
            cur_goal[b, :-1] = rs.randint(3, 32000, size=L - 1)
        cur_t = cur_t + 1
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(spec.seed)
    st = RolloutStorage(T, device=dev, store_tokens=True)
    st.T, st.B = T, B
    f = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
    st.observations = {
        "dino_tokens": torch.randn(T + 1, B, 2, 84, 384, device=dev, dtype=torch.float32, generator=g).to(torch.bfloat16),
        "goal_token_ids": torch.from_numpy(goal).to(dev),
        "time_step": torch.from_numpy(time_step).to(dev),
        "traj_index": torch.from_numpy(traj).to(dev),
        "an_object_is_in_hand": torch.from_numpy((rs.rand(T + 1, B, 1) < (0.0 if spec.task == "ObjectNav" else 0.1)).astype(np.int64)).to(dev),
    }
    st.masks = torch.from_numpy(masks).to(dev)
    st.rewards, st.costs = torch.from_numpy(rewards).to(dev), torch.from_numpy(costs).to(dev)
    st.prev_actions = torch.from_numpy(rs.randint(0, 20, size=(T + 1, B)).astype(np.int64)).to(dev)
    st.value_preds, st.c_value_preds = f(T + 1, B, 1), f(T + 1, B, 1)
    st.returns, st.c_returns = f(T + 1, B, 1), f(T + 1, B, 1)
    st.step = T
    # "acting" products: old log-probs / values from a real no-grad pass of the same policy (ratios start at 1)
    with torch.no_grad():
        obs = {k: v[:T] for k, v in st.observations.items()}
        out, _ = model(obs, None, st.prev_actions[:T], st.masks[:T])
        st.actions = out.distributions.sample()
        st.action_log_probs = out.distributions.log_prob(st.actions).float().contiguous()
        st.value_preds[:T] = out.values
        st.c_value_preds[:T] = out.c_values
        last = {k: v[T:T + 1] for k, v in st.observations.items()}
        o2, _ = model(last, None, st.prev_actions[T:T + 1], st.masks[T:T + 1])
        nxt = dict(next_value=o2.values.reshape(B, 1).clone(), next_c_value=o2.c_values.reshape(B, 1).clone())
    return st, nxt, dict(episode_cost_sum=float(ep_cost_sum), n_episodes=float(n_ep))
