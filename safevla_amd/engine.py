"""PPO-Lagrangian update engine (single- and multi-GPU).

Replaces the update half of the AllenAct-fork ``OnPolicyTrainer`` [3P, not in /root/reference] as configured by the
reference's ``training_pipeline()`` (/root/reference/training/online/dinov2_vits_tsfm_base.py:293-380):
Adam(lr), ``num_mini_batch=1``, ``update_repeats=4``, ``max_grad_norm=0.5``, gamma 0.99, GAE lambda 0.95, stage loss
lists, ``cost_limit`` via ``start_train(cost_limit=...)`` (/root/reference/training/online/allenact_trainer.py:63-72).

One update = GAE(reward+cost) -> lambda update -> update_repeats x num_mini_batch x
   [zero grads; per env-chunk: 3 x (tower forward, fused loss fwd+bwd, tower backward); all-reduce flat grads;
    global-norm clip + Adam (one fused launch over the flat arena); refresh bf16 transposes].
Towers run one after another (each tower's loss depends only on its own outputs), so only one tower's activations are
resident at a time; env-chunking gives exact gradient accumulation because every loss is a mean over rows
(inv_n = 1 / global rows).

Data parallel: each tower owns one contiguous, aligned range of the flat gradient arena; as soon as a tower's backward
of the LAST env-chunk has been issued its range is all-reduced asynchronously (RCCL runs on its own stream), overlapping the
next tower's forward + backward; the optimiser step waits for the three handles.  Host syncs per update: ONE at the very start
(the [sum cost, #episodes] all-reduce feeding the host-side lambda, before any kernel of the update is queued) and ONE at the end
(the info scalars); the global row count of a minibatch is reduced once per storage geometry and cached.

Adam bookkeeping follows torch.optim.Adam under ``zero_grad(set_to_none=True)`` (what the reference engine runs): a tower whose
loss is not in the current stage's loss list has ``grad is None`` for all of its parameters, so the optimiser skips it -- its
moments freeze and its per-parameter step count does not advance.  Here: per-tower step counters, inactive ranges are skipped.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import weakref

import torch

from . import ops, parallel
from .lagrange import Lagrange
from .model import N_ACTIONS, SafeDinoLLAMATxNavActorCriticSeparate
from .storage import RolloutStorage


@dataclass
class PPOLagConfig:
    update_repeats: int = 4
    num_mini_batch: int = 1
    lr: float = 2e-5
    max_grad_norm: float = 0.5
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_param: float = 0.1
    value_loss_coef: float = 0.5
    entropy_coef: float = 0.0
    action_weight: float = 1.0
    stage_losses: Tuple[str, ...] = ("ppo_log_loss", "safe_ppo_value_loss")
    cost_limit: float = 2.31964          # README.md:255 example
    lambda_init: float = 0.001
    lambda_lr: float = 0.035
    lambda_optimizer: str = "Adam"
    lambda_upper_bound: Optional[float] = None
    env_chunk: Optional[int] = None       # envs per micro-batch (None: whole local minibatch)
    record_small_updates: bool = True     # small minibatches: record each env-chunk's launch sequence in the first epoch, replay it in the others
    adam_betas: Tuple[float, float] = (0.9, 0.999)
    adam_eps: float = 1e-8
    deterministic: bool = False           # bitwise-repeatable gradients: cross-workgroup accumulation in 64-bit fixed point (svla_det_config)
    grad_allreduce_dtype: str = "fp32"    # "bf16": the per-tower gradient ranges cross xGMI as bf16 (126 MB instead of 252 MB per step; parallel._Bf16Exchange)


class PPOLagEngine:
    def __init__(self, model: SafeDinoLLAMATxNavActorCriticSeparate, cfg: PPOLagConfig):
        self.model, self.cfg = model, cfg
        self.lagrange = Lagrange(cfg.cost_limit, cfg.lambda_init, cfg.lambda_lr, cfg.lambda_optimizer, cfg.lambda_upper_bound)
        self.opt_step = 0                     # optimiser steps taken (any tower)
        self.tower_steps = [0, 0, 0]          # torch.optim.Adam's per-parameter ``step`` (identical within a tower)
        self._pending = []                    # async all-reduce handles of the current minibatch
        self._chunk_cache = {}                # small minibatches: recorded launch sequences per env-chunk, valid within one update
        dev = model.device_
        self._gnorm_sq = torch.zeros(1, device=dev, dtype=torch.float64)
        self._sums = torch.zeros(5, device=dev, dtype=torch.float64)   # v_sq, action, -entropy, hl-gauss CE (discrete critic), c_v_sq
        self.gemm_flops = 0
        # deterministic mode: an int64 shadow of the flat gradient buffer (8 B per parameter)
        self._det_shadow = torch.zeros(model.arena.flat_g.numel(), device=dev, dtype=torch.int64) if cfg.deterministic else None
        if not hasattr(model, "_invalidate_hooks"):
            model._invalidate_hooks = []
        # load_state_dict / broadcast replace the frozen encoder's tensors: recorded env-chunks must go.  The hook holds the engine weakly --
        # a strong reference to the cache would keep a discarded engine's recorded chunks (and their towers' activations) alive
        ref = weakref.ref(self)
        model._invalidate_hooks[:] = [h for h in model._invalidate_hooks if getattr(h, "_alive", lambda: True)()]
        hook = lambda: (ref() is not None and ref()._chunk_cache.clear())
        hook._alive = lambda: ref() is not None
        model._invalidate_hooks.append(hook)
        if parallel.is_dist():      # independent dropout noise per rank (every rank holds different environments)
            for t in model.towers:
                t.drop_seed_base += 7919 * torch.distributed.get_rank()

    # ---- one minibatch: forward/backward of the three towers with fused losses ------------------------------------
    def active_towers(self) -> Tuple[bool, bool, bool]:
        """(actor, reward critic, cost critic): which towers receive a gradient under the current stage's loss list."""
        names = set(self.cfg.stage_losses)
        discrete = self.model.critic_type == "discrete"
        return ("ppo_log_loss" in names, ("ppo_log_loss" in names and not discrete) or "ppo_value_loss" in names,
                "safe_ppo_value_loss" in names or ("ppo_log_loss" in names and discrete))

    def _reduce_tower_async(self, k: int):
        if parallel.is_dist():
            a, b = self.model.arena.tower_ranges[k]
            wire = torch.bfloat16 if self.cfg.grad_allreduce_dtype == "bf16" else None
            self._pending.append(parallel.allreduce_sum_async(self.model.arena.flat_g[a:b], wire_dtype=wire))

    def _accumulate(self, batch: Dict, n_total: int, lam: float, last: bool = False, cache_key=None):
        """``last``: this is the final env-chunk of the minibatch -- each tower's gradient range is handed to the asynchronous
        all-reduce right after that tower's backward has been issued.

        ``cache_key`` (small minibatches only): the caller promises that this chunk -- same storage contents, lambda, stage -- comes
        back under the same key (the epochs of one update).  The first call records each tower's launch sequence (forward, fused loss,
        backward) as an ``ops.LaunchPlan`` while executing it; later calls re-issue the three lists on the three tower streams with
        fresh device-resident dropout seeds.  ~1000 small dependent launches per 3-tower pass cost ~14 us each through the Python
        wrappers and ~6.5 us replayed (tools/replay_probe.py)."""
        cfg, m = self.cfg, self.model
        T, Bc = batch["actions"].shape
        R = T * Bc
        inv_n = 1.0 / float(n_total)
        names = set(cfg.stage_losses)
        st = self._chunk_cache.get(cache_key) if cache_key is not None else None
        if st is not None and st["sig"] == (R, n_total, lam, tuple(cfg.stage_losses), m.training):
            main = torch.cuda.current_stream()
            for k, plan in enumerate(st["plans"]):
                if plan is None:
                    continue
                s_ = m._tower_streams[k]
                m.towers[k]._seed_dev_buf.add_(0x3C6EF35)          # fresh dropout noise for this pass (forward and backward read the same seed)
                s_.wait_stream(main)
                plan.replay()
                if last and parallel.is_dist():
                    with torch.cuda.stream(s_):
                        self._reduce_tower_async(k)
            for s_ in m._tower_streams:
                main.wait_stream(s_)
            return
        prep = m.prepare(batch["observations"], batch["prev_actions"], batch["masks"])
        f = lambda t: t.reshape(R).contiguous()
        sums = self._sums
        ret_ = f(batch["returns"])
        discrete = m.critic_type == "discrete"
        small = m.concurrent_towers and R * prep.S <= m.concurrent_tower_tokens
        record = (small and cache_key is not None and cfg.record_small_updates and m.adt == torch.bfloat16 and m.critic_type == "linear"
                  and not cfg.deterministic        # the accumulation mode is read at launch time, not part of a recorded launch
                  and len(self._chunk_cache) < 4)        # a recorded chunk keeps its three towers' activations alive: bound the footprint
        flat = {k: f(batch[k]) for k in ("actions", "old_action_log_probs", "adv_targ", "c_adv_targ", "c_returns")}     # contiguous once, outside the recorded region

        def actor_block():
            # actor: clipped surrogate on the lambda-mixed advantage (+ entropy); critic: value_loss_coef * 0.5 * mse
            logits, _, c = m.run_forward(prep, need_grad=True)
            _, dl, _ = ops.ppo_lag_loss_fwd_bwd(logits.reshape(R, N_ACTIONS), ret_, flat["actions"], flat["old_action_log_probs"],
                                                flat["adv_targ"], flat["c_adv_targ"], ret_, None, lam, cfg.clip_param, 0.0,
                                                cfg.action_weight, cfg.entropy_coef, False, inv_n, sums=sums[0:3])
            m.run_backward(prep, c, dl.view(T, Bc, N_ACTIONS), None)

        def critic_block():
            coef = cfg.value_loss_coef if "ppo_log_loss" in names else 1.0
            tw = m.critic_tsfm
            _, values, c = tw.run_forward(prep, need_grad=True)
            _, dv = ops.value_mse_fwd_bwd(values.reshape(R), ret_, coef, inv_n, sums=sums[0:1])
            tw.run_backward(prep, c, None, dv.view(T, Bc, 1))

        def c_critic_block():
            tw = m.c_critic_tsfm
            _, c_values, c = tw.run_forward(prep, need_grad=True)
            dv = dfl = None
            if "safe_ppo_value_loss" in names:
                _, dv = ops.value_mse_fwd_bwd(c_values.reshape(R), flat["c_returns"], 1.0, inv_n, sums=sums[4:5])
                dv = dv.view(T, Bc, 1)
            if "ppo_log_loss" in names and discrete:
                # the reference's own data flow with critic_type="discrete": SafePPOLogGrad's value term is HL-Gauss(extras["full_logits"],
                # returns) and the 3-tower wrapper's extras are the COST-critic tower's (customized_loss.py:364-370, separate_actor_critic.py:31-36)
                lf, fl = tw.critic.loss_fn, tw._last_full_logits
                _, dfl, _ = ops.hlgauss_fwd_bwd(fl.reshape(R, fl.shape[-1]), ret_, None, lf.min_value, lf.max_value, lf.sigma, 0.5 * cfg.value_loss_coef, inv_n,
                                                want_values=False, sums=sums[3:4])
                dfl = dfl.view(T, Bc, -1)
            tw.run_backward(prep, c, None, dv, dfl)

        blocks = [actor_block if "ppo_log_loss" in names else None,
                  critic_block if (("ppo_log_loss" in names and not discrete) or "ppo_value_loss" in names) else None,
                  c_critic_block if ("safe_ppo_value_loss" in names or ("ppo_log_loss" in names and discrete)) else None]
        def run_block(k, t):
            if blocks[k] is None:
                return None
            plan = None
            if record:
                if getattr(t, "_seed_dev_buf", None) is None:
                    t._seed_dev_buf = torch.tensor([(t.drop_seed_base * 0x9E3779B1) & 0x7FFFFFFF], device=m.device_, dtype=torch.int32)
                t._seed_dev_buf.add_(0x3C6EF35)
                t._seed_dev = t._seed_dev_buf          # dropout descriptors of the recorded pass point at the device-resident seed
                plan = ops.LaunchPlan()
                try:
                    with plan:
                        blocks[k]()
                finally:
                    t._seed_dev = None
            else:
                blocks[k]()
            if last:
                if cfg.deterministic:      # fold this tower's fixed-point sums into its fp32 range before anything reads it
                    a, b = m.arena.tower_ranges[k]
                    ops.det_finalize(m.arena.flat_g[a:b], self._det_shadow[a:b])
                self._reduce_tower_async(k)
            return plan

        if cfg.deterministic:
            ops.det_set_grid(ops.det_grid_bits(n_total))      # (shadows are empty here: every range is folded back at the end of its tower's pass)
            ops.det_config(0, m.arena.flat_g, self._det_shadow)
        try:
            if small:
                # small minibatches are bound by the dispatch of ~1000 small dependent kernels: the three towers (independent given the batch)
                # run on three HIP streams (model.run_towers_concurrently); gradients land in disjoint arena ranges, loss sums are atomics
                plans = m.run_towers_concurrently(run_block)
                if record:
                    self._chunk_cache[cache_key] = dict(sig=(R, n_total, lam, tuple(cfg.stage_losses), m.training), plans=plans, keep=(prep, flat, ret_))
            else:
                for k, t in enumerate(m.towers):      # update-sized shapes: one tower at a time (only one tower's activations resident)
                    run_block(k, t)
        finally:
            if cfg.deterministic:
                ops.det_config(0, None, None)

    def optimizer_step(self, reduced: bool = False):
        """Global-norm clip + Adam over the ranges of the towers that received a gradient.  ``reduced``: the per-tower asynchronous
        all-reduces were already started by ``_accumulate(last=True)`` (wait for them); otherwise reduce the active ranges here."""
        cfg, ar = self.cfg, self.model.arena
        active = self.active_towers()
        if reduced:
            for w in self._pending:
                w.wait()
        elif parallel.is_dist():
            for k, on in enumerate(active):
                if on:
                    a, b = ar.tower_ranges[k]
                    if cfg.grad_allreduce_dtype == "bf16":
                        parallel.allreduce_sum_async(ar.flat_g[a:b], wire_dtype=torch.bfloat16).wait()
                    else:
                        parallel.allreduce_sum_(ar.flat_g[a:b])
        self._pending = []
        self._gnorm_sq.zero_()
        for k, on in enumerate(active):       # clip_grad_norm_ sees the parameters that have a gradient
            if on:
                a, b = ar.tower_ranges[k]
                ops.sumsq(ar.flat_g[a:b], self._gnorm_sq)
        self.opt_step += 1
        for k, on in enumerate(active):
            if not on:
                continue
            a, b = ar.tower_ranges[k]
            self.tower_steps[k] += 1
            ops.adam_step(ar.flat_p[a:b], ar.flat_g[a:b], ar.flat_m[a:b], ar.flat_v[a:b], ar.flat_bf16[a:b], cfg.lr, self.tower_steps[k],
                          cfg.adam_betas[0], cfg.adam_betas[1], cfg.adam_eps, gnorm_sq=self._gnorm_sq, max_norm=cfg.max_grad_norm)
            self.model.towers[k].refresh_transposes()

    def _global_rows(self, local_rows, dev):
        """Global row counts of a list of local counts (every minibatch of the update + the whole rollout): ONE all-reduce, issued by every
        rank at the same point of update() whatever its shard size -- env shards may be uneven (parallel.shard_envs), so a per-value cache
        would make ranks issue different numbers of collectives."""
        if not parallel.is_dist():
            return [int(r) for r in local_rows]
        return parallel.global_counts(local_rows, dev)

    # ---- one full PPO-Lagrangian update on a filled storage --------------------------------------------------------------
    def update(self, storage: RolloutStorage, next_value: torch.Tensor, next_c_value: torch.Tensor,
               episode_cost_sum: float, n_episodes: float, generator: Optional[torch.Generator] = None) -> Dict[str, float]:
        cfg, m = self.cfg, self.model
        dev = m.device_
        if not parallel.is_dist() and cfg.num_mini_batch > storage.B:       # refused before anything (returns, multiplier) is touched; the data-parallel form of the check follows the count all-reduce
            raise ValueError(f"num_mini_batch = {cfg.num_mini_batch} leaves a minibatch without environments ({storage.B} env(s))")
        self._chunk_cache.clear()             # a new rollout: recorded sequences of the previous update refer to other inputs
        storage.compute_returns(next_value, next_c_value, True, cfg.gamma, cfg.gae_lambda)
        Jc, n_ep = parallel.mean_episode_cost(episode_cost_sum, n_episodes, dev)
        lam = self.lagrange.update_lagrange_multiplier(Jc) if n_ep > 0 else self.lagrange.lagrangian_multiplier
        T, B = storage.T, storage.B
        info_acc = torch.zeros(5, device=dev, dtype=torch.float64)
        n_mb = 0
        bounds = [round(i * B / cfg.num_mini_batch) for i in range(cfg.num_mini_batch + 1)]
        counts = self._global_rows([T * (bounds[i + 1] - bounds[i]) for i in range(cfg.num_mini_batch)] + [T * B], dev)
        if min(counts[:-1]) <= 0:
            # more minibatches than environments on EVERY rank together: a minibatch without a row has no gradient and no loss (AllenAct refuses the same
            # configuration when it builds its samplers); a rank-local empty minibatch under data parallelism is legal and handled below
            raise ValueError(f"num_mini_batch = {cfg.num_mini_batch} leaves a minibatch without environments ({B} local env(s); global rows per minibatch {counts[:-1]})")
        if parallel.is_dist() and cfg.num_mini_batch > 1:
            # the i-th minibatch of every rank forms ONE global minibatch: all ranks must walk them in the same order
            generator = torch.Generator().manual_seed(0x5AFE + self.opt_step)
        for _ in range(cfg.update_repeats):
            order = torch.randperm(cfg.num_mini_batch, generator=generator).tolist() if cfg.num_mini_batch > 1 else [0]
            for i in order:
                b0, b1 = bounds[i], bounds[i + 1]
                n_total = counts[i]
                m.zero_grad()
                self._sums.zero_()
                chunk = cfg.env_chunk or max(1, b1 - b0)
                for c0 in range(b0, b1, chunk):
                    self._accumulate(storage.batch_slice(c0, min(b1, c0 + chunk)), n_total, lam, last=c0 + chunk >= b1,
                                     cache_key=(c0, min(b1, c0 + chunk)))
                # a rank whose shard has fewer envs than minibatches (uneven strong-scaling shards) owns NO rows of this minibatch: it still joins the
                # three per-tower all-reduces (in tower order, like the asynchronous ones of the other ranks) with its zeroed gradient
                self.optimizer_step(reduced=b1 > b0)
                parallel.allreduce_sum_(self._sums)
                info_acc += self._sums / n_total
                n_mb += 1
        s = (info_acc / n_mb).cpu().tolist()      # host sync at the end of the update (the other one is the Jc reduction at its start)
        value, action, ent, c_value = 0.5 * (s[3] if m.critic_type == "discrete" else s[0]), s[1], s[2], 0.5 * s[4]
        info = {"ppo_total": cfg.value_loss_coef * value + cfg.action_weight * action + cfg.entropy_coef * ent, "value": value,
                "action": action, "entropy": ent, "c_value": c_value, "lagrangian_multiplier": lam, "Jc": Jc,
                "env_steps": counts[-1]}
        if cfg.deterministic:
            # partial sums that left the fixed-point shadow (|partial| >= 0.25, NaN, Inf: plain fp32 atomics): the update was bitwise repeatable iff 0
            info["det_bypassed_partials"] = ops.det_bypass_count(reset=True)
        return info
