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

from .api import SafeRLStepResult
from .storage import RolloutStorage

TASK_DONE_P = {"ObjectNav": 1.0 / 60, "PickUp": 1.0 / 42, "Fetch": 1.0 / 110}
TASK_HAND_P = {"ObjectNav": 0.0, "PickUp": 0.1, "Fetch": 0.1}
MIXED_ORDER = ("ObjectNav", "PickUp", "Fetch")


def env_tasks(task: str, B: int, env_offset: int = 0):
    """Task type of every local environment.  ``task="Mixed"`` = BASELINE configs[4]'s multi-task sampler as SURVEY 8(d) scopes it:
    (global) env e runs task e mod 3 of ObjectNav / PickUp / Fetch (the reference mixes task types per sampler process,
    /root/reference/tasks/multi_task_eval_sampler.py, tasks/task_specs.py); ``env_offset`` = index of this rank's first env."""
    if task == "Mixed":
        return [MIXED_ORDER[(env_offset + b) % 3] for b in range(B)]
    return [task] * B


@dataclass
class SynthSpec:
    T: int = 256
    B: int = 32
    L: int = 12                 # goal tokens (constant per episode)
    task: str = "Fetch"
    seed: int = 1234
    max_steps: int = 500
    cost_p: float = 0.05        # cost ~ Binomial(5, cost_p)
    env_offset: int = 0         # global index of the first local env (task assignment of the mixed sampler under DP)


def fill_synthetic_rollout(model, spec: SynthSpec, device="cuda") -> Tuple[RolloutStorage, Dict[str, torch.Tensor], Dict[str, float]]:
    """Returns (storage filled with T steps, {next_value, next_c_value}, {episode_cost_sum, n_episodes})."""
    T, B, L = spec.T, spec.B, spec.L
    rs = np.random.RandomState(spec.seed)
    tasks = env_tasks(spec.task, B, spec.env_offset)
    done_p = np.array([TASK_DONE_P.get(t, 1.0 / 60) for t in tasks])
    hand_p = np.array([TASK_HAND_P.get(t, 0.1) for t in tasks])
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
        "an_object_is_in_hand": torch.from_numpy((rs.rand(T + 1, B, 1) < hand_p[None, :, None]).astype(np.int64)).to(dev),
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


class SynthVectorEnv:
    """Steppable stand-in for AllenAct's ``VectorSampledTasks`` over AI2-THOR: B independent environments living in device memory,
    ``reset() -> observations`` and ``step(actions) -> (observations, rewards, costs, dones, results)`` with the sensor contracts of
    SURVEY Appendix D (two pre-encoded DINOv2 feature maps, goal token ids, ``time_step``, ``traj_index`` mod 2048,
    ``an_object_is_in_hand``), reward +10 on a successful ``end`` and the 0..5 integer safety cost of
    /root/reference/tasks/abstract_task.py:321-333.  ``results`` (on request) are per-env ``SafeRLStepResult``s -- what
    ``Task.step`` returns in the reference (abstract_task.py:369-381).  Episodes end when the policy emits ``end`` (action 4) with
    the task's success statistics, by the per-task hazard rate, or at ``max_steps``."""

    END_ACTION = 4          # ALL_STRETCH_ACTIONS[4] == "end" (utils/constants/stretch_initialization_utils.py:145-166)

    def __init__(self, B: int, L: int = 12, task: str = "ObjectNav", seed: int = 0, max_steps: int = 500, cost_p: float = 0.05,
                 device="cuda", env_offset: int = 0, store_tokens: bool = True):
        self.B, self.L, self.max_steps, self.cost_p = B, L, max_steps, cost_p
        self.dev = torch.device(device)
        self.g = torch.Generator(device=self.dev).manual_seed(seed)
        self.tasks = env_tasks(task, B, env_offset)
        self.done_p = torch.tensor([TASK_DONE_P.get(t, 1.0 / 60) for t in self.tasks], device=self.dev)
        self.hand_p = torch.tensor([TASK_HAND_P.get(t, 0.1) for t in self.tasks], device=self.dev)
        self.store_tokens = store_tokens
        self.time_step = torch.zeros(B, device=self.dev, dtype=torch.int64)
        self.traj = torch.randint(0, 2048, (B,), device=self.dev, generator=self.g)
        self.goal = self._new_goals(B)
        self.ep_cost = torch.zeros(B, device=self.dev)
        self.finished_cost_sum = torch.zeros((), device=self.dev, dtype=torch.float64)
        self.finished_episodes = torch.zeros((), device=self.dev, dtype=torch.float64)

    def _new_goals(self, n):
        g = torch.randint(3, 32000, (n, self.L), device=self.dev, generator=self.g)
        g[:, -1] = 1       # EOS
        return g

    def _rand(self, *shape):
        return torch.rand(*shape, device=self.dev, generator=self.g)

    def observations(self) -> Dict[str, torch.Tensor]:
        B = self.B
        obs = {"goal_token_ids": self.goal.clone(), "time_step": self.time_step.clone(), "traj_index": self.traj.clone(),
               "an_object_is_in_hand": (self._rand(B, 1) < self.hand_p[:, None]).to(torch.int64)}
        if self.store_tokens:
            obs["dino_tokens"] = torch.randn(B, 2, 84, 384, device=self.dev, generator=self.g).to(torch.bfloat16)
        else:
            obs["rgb_dinov2"] = torch.randn(B, 384, 7, 12, device=self.dev, generator=self.g)
            obs["manipulation_rgb_dinov2"] = torch.randn(B, 384, 7, 12, device=self.dev, generator=self.g)
        return obs

    def reset(self) -> Dict[str, torch.Tensor]:
        self.time_step.zero_()
        return self.observations()

    def step(self, actions: torch.Tensor, want_results: bool = False):
        B = self.B
        a = actions.reshape(B)
        cost = torch.binomial(torch.full((B,), 5.0, device=self.dev), torch.full((B,), self.cost_p, device=self.dev), generator=self.g)
        ended = a == self.END_ACTION
        done = ended | (self._rand(B) < self.done_p) | (self.time_step + 1 >= self.max_steps)
        reward = 10.0 * (done & (self._rand(B) < 0.5)).float()
        self.ep_cost += cost
        self.finished_cost_sum += (self.ep_cost * done).sum().double()
        self.finished_episodes += done.sum().double()
        self.ep_cost = torch.where(done, torch.zeros_like(self.ep_cost), self.ep_cost)
        self.time_step = torch.where(done, torch.zeros_like(self.time_step), self.time_step + 1)
        self.traj = torch.where(done, (self.traj + 1) % 2048, self.traj)
        self.goal = torch.where(done[:, None], self._new_goals(B), self.goal)
        obs = self.observations()
        results = None
        if want_results:     # host-side view, one SafeRLStepResult per env (diagnostics / API parity; the training loop uses the tensors)
            r, c, d = reward.cpu().tolist(), cost.cpu().tolist(), done.cpu().tolist()
            results = [SafeRLStepResult(observation={k: v[b] for k, v in obs.items()}, reward=r[b], cost=c[b], done=bool(d[b]),
                                        info={"action": int(a[b]), "task_type": self.tasks[b]}) for b in range(B)]
        return obs, reward, cost, done, results

    def pop_episode_costs(self):
        """[sum of finished-episode costs, number of finished episodes] since the last call (device scalars -> one host copy)."""
        s, n = float(self.finished_cost_sum), float(self.finished_episodes)
        self.finished_cost_sum.zero_(); self.finished_episodes.zero_()
        return s, n


@torch.no_grad()
def collect_rollout(model, env: SynthVectorEnv, storage: RolloutStorage, T: int, obs0: Optional[Dict[str, torch.Tensor]] = None):
    """One rollout through the ACTING path, the way the reference engine collects experience: per step a single-step
    3-tower forward with the llama KV caches, a sample from the policy, ``env.step``, ``storage.add``.  Returns the bootstrap values."""
    B = env.B
    if obs0 is not None:
        storage.initialize(obs0, num_samplers=B)
    for _ in range(T):
        inp = storage.agent_input_for_next_step()
        out, _ = model(inp["observations"], None, inp["prev_actions"], inp["masks"])
        actions = out.distributions.sample().reshape(B)
        logp = out.distributions.log_prob(actions.reshape(1, B)).reshape(B)
        obs, reward, cost, done, _ = env.step(actions)
        storage.add(obs, None, actions, logp, out.values.reshape(B, 1), reward.reshape(B, 1), cost.reshape(B, 1),
                    out.c_values.reshape(B, 1), (1.0 - done.float()).reshape(B, 1))
    inp = storage.agent_input_for_next_step()
    for t in model.towers:                      # bootstrap forward must not advance the caches / counters
        t._saved_counter = t.time_step_counter
    out, _ = model(inp["observations"], None, inp["prev_actions"], inp["masks"])
    for t in model.towers:
        t.time_step_counter = t._saved_counter
    return dict(next_value=out.values.reshape(B, 1).clone(), next_c_value=out.c_values.reshape(B, 1).clone())
