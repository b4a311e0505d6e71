#!/usr/bin/env python3
"""bench.py -- env-steps/s through the PPO-Lagrangian update on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = ONE full PPO-Lagrangian update over a synthetic rollout already resident in HBM:
  reward+cost GAE  ->  lambda update  ->  update_repeats(4) x [3 towers x (forward, fused loss, backward),
  all-reduce of the flat gradient arena, global-norm clip + Adam].
Workload at N=1 = one GPU's shard of BASELINE.json configs[3] ("Fetch, 256 envs sharded 8xMI355X, 256-step rollout"):
T=256 steps x 32 envs per GPU, 12 goal tokens; weak scaling (32 envs per GPU at every N).
Prints ONE JSON line (rank 0) with the roofline of the dominant kernel (the bf16 MFMA GEMM, timed with HIP events on
its launch stream in a separate instrumented update) and a CPU baseline (the fp32 oracle port on the host cores, bounded
sample).  The oracle is only the checker/baseline leg here -- the measured path is the HIP library.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def flops_per_update(R, S, L, U, epochs):
    """Algorithmic FLOPs (2*MAC) of the trainable path, SURVEY.md section 8(d): per row per tower forward,
    x3 (fwd + 2x bwd) x 3 towers x epochs.  Frozen T5 (once per unique goal) excluded."""
    d = 512
    fusion = 3 * S * (4 * d * d + 2 * d * 2048 + 2 * S * d) * 2
    compress = 2 * 84 * (384 * d + d * d + d * d) * 2
    text = L * d * d * 2
    decoder = 3 * (4 * d * d + 3 * d * 1536 + 2 * 256 * d) * 2 + d * d * 2 + 21 * d * 2
    return R * 3 * 3 * epochs * (fusion + compress + decoder) + U * 3 * 3 * epochs * text


class GemmTimer:
    """HIP-event timing of every MFMA-kernel launch (on torch's current stream = the launch stream) during one update."""

    def __init__(self, ops):
        self.ops, self.rec = ops, {"gemm_nt256": [], "gemm_nt": [], "gemm_tn": [], "attn_fwd": [], "attn_bwd": []}
        self.orig = {"gemm_nt": ops.gemm_nt, "gemm_tn": ops.gemm_tn_acc, "attn_fwd": ops.attn_fwd, "attn_bwd": ops.attn_bwd}

    def _wrap(self, key, fn, flops):
        def wrapped(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            k2 = key
            if key == "gemm_nt":   # same dispatch rule as svla_gemm_nt_bf16: big row-streaming shapes run the persistent 256x256 kernel
                M, N, K = a[2], a[3], a[4]
                if not kw.get("out_f32") and N % 256 == 0 and K % 64 == 0 and K >= 128 and ((M + 255) // 256) * (N // 256) >= 256:
                    k2 = "gemm_nt256"
            self.rec[k2].append((e0, e1, flops(*a, **kw)))
            return out

        return wrapped

    def __enter__(self):
        o = self.ops
        o.gemm_nt = self._wrap("gemm_nt", self.orig["gemm_nt"], lambda A, B, M, N, K, **kw: 2.0 * M * N * K)
        o.gemm_tn_acc = self._wrap("gemm_tn", self.orig["gemm_tn"], lambda dY, X, dW, M, N, K, **kw: 2.0 * M * N * K)
        o.attn_fwd = self._wrap("attn_fwd", self.orig["attn_fwd"],
                                lambda q, k, v, ld, rows, S, H, scale, **kw: 4.0 * (kw.get("Sq") or S) * S * 64 * rows * H)
        o.attn_bwd = self._wrap("attn_bwd", self.orig["attn_bwd"],
                                lambda q, k, v, ld, o_, ldo, lse, do, lddo, dq, dk, dv, ldd, rows, S, H, scale, **kw:
                                10.0 * (kw.get("Sq") or S) * S * 64 * rows * H)
        return self

    def __exit__(self, *a):
        o = self.ops
        o.gemm_nt, o.gemm_tn_acc, o.attn_fwd, o.attn_bwd = (self.orig[k] for k in ("gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd"))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, rec in self.rec.items():
            ms = [e0.elapsed_time(e1) for e0, e1, _ in rec]
            fl = sum(f for _, _, f in rec)
            tot = sum(ms) * 1e-3
            out[key] = dict(launches=len(ms), avg_ms=sum(ms) / max(1, len(ms)), tflops=fl / max(tot, 1e-12) / 1e12, flops=fl, total_s=tot)
        return out


def acting_bench(model, st, B, dev, n=24):
    """Secondary number ("report separately the rollout-time throughput"): the acting path that precedes the update --
    frozen DINOv2 ViT on 2 uint8 frames per env step + single-step 3-tower forward with the llama KV caches."""
    from safevla_amd.preproc import DinoViTPreprocessor

    step_in = lambda t: ({k: v[t:t + 1] for k, v in st.observations.items()}, None, st.prev_actions[t:t + 1], st.masks[t:t + 1])

    def policy_rate(graph):
        for t in model.towers:
            t.time_step_counter, t._kv = 0, None
        model.enable_acting_graphs(graph)
        with torch.no_grad():
            for t in range(4):
                model(*step_in(t))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(4, 4 + n):
                model(*step_in(t))
            torch.cuda.synchronize()
            r = n * B / (time.perf_counter() - t0)
        model.enable_acting_graphs(False)
        for t in model.towers:
            t.time_step_counter, t._kv = 0, None
        return r

    pol_eager, pol = policy_rate(False), policy_rate(True)
    vit = DinoViTPreprocessor("rgb_raw", "rgb_dinov2", device=dev)
    fr = torch.randint(0, 256, (2 * B, 224, 384, 3), device=dev, dtype=torch.uint8)
    vit.process({"rgb_raw": fr})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        vit.process({"rgb_raw": fr})
    torch.cuda.synchronize()
    fps = 3 * 2 * B / (time.perf_counter() - t0)
    best = max(pol, pol_eager)
    return {"policy_single_step_env_steps_per_s": round(pol_eager, 1), "policy_single_step_hipgraph_env_steps_per_s": round(pol, 1),
            "vit_frames_per_s": round(fps, 1),
            "acting_env_steps_per_s": round(1.0 / (1.0 / best + 2.0 / fps), 1), "envs": B,
            "note": "per env step: 2 frames (224x384) through DINOv2 ViT-S/14 + one KV-cached 3-tower step; eager = one Python-issued launch "
                    "per kernel, hipgraph = the same step captured once and replayed (model.enable_acting_graphs); synthetic frames"}


def north_star_probe(model, dev, rows_T=32, rows_B=8, L=64):
    """Secondary number named by BASELINE.json's north_star: MFMA-roofline fraction of the policy forward+backward at batch 256
    with a 64-token instruction (3 towers, fused losses, no optimiser step; pre-encoded features -- the frozen ViT is not part of
    the reference's update path).  Algorithmic FLOPs per SURVEY 8(d) / measured time / 2.5 PFLOP/s."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=rows_T, B=rows_B, L=L, task="Fetch", seed=99), device=dev)
    eng = PPOLagEngine(model, PPOLagConfig())
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    batch = st.batch_slice(0, rows_B)
    R = rows_T * rows_B

    def once():
        model.zero_grad()
        eng._accumulate(batch, R, 0.1)

    once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    U = int(st.observations["goal_token_ids"][:rows_T].reshape(R, -1).unique(dim=0).shape[0])
    fl = flops_per_update(R, 169 + L, L, U, 1)
    return {"rows": R, "goal_tokens": L, "ms_fwd_bwd_3_towers": round(ms, 2), "algorithmic_tflop": round(fl / 1e12, 2),
            "achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 1), "frac_of_bf16_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "note": "north_star secondary target (>= 0.70) -- small batch: 256 rows x 233 tokens"}


def cpu_baseline(T=32, B=8, L=12, train_mode=True, threads=None):
    """fp32 CPU port (oracle) of the same update on a bounded sample: one epoch = 3-tower forward, SafePPOLogGrad +
    SafePPOValue, backward, clip 0.5, Adam; env-steps/s = T*B / (4 epochs)."""
    import numpy as np

    from oracle import ref_loss, ref_model
    from safevla_amd.text import GoalTokenizer

    n = threads or min(16, os.cpu_count() or 1)   # more threads only add fork/join overhead on these op sizes (measured: 256 threads 40x slower)
    torch.set_num_threads(n)
    torch.manual_seed(0)

    class _Tok:  # fixed-length synthetic ids, same as the GPU workload
        def __call__(self, goals, return_tensors="pt", padding=True):
            ids = torch.randint(3, 32000, (len(goals), L))
            return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}

    m = ref_model.RefSafeActorCritic(_Tok(), max_steps=500, max_batch=B, dropout=0.1 if train_mode else 0.0)
    m.train(train_mode)       # the same mode as the GPU leg (the reference trains with dropout 0.1 on); T5 stays deterministic
    for t in (m, m.critic_tsfm, m.c_critic_tsfm):
        t.visual_encoder.text_encoder.eval()
    params = [p for n_, p in m.named_parameters() if "text_encoder" not in n_]
    opt = torch.optim.Adam(params, lr=2e-5)
    obs = {"rgb_dinov2": torch.randn(T, B, 384, 7, 12), "manipulation_rgb_dinov2": torch.randn(T, B, 384, 7, 12),
           "natural_language_spec": torch.zeros(T, B, 1000, dtype=torch.uint8), "time_step": torch.arange(T)[:, None].expand(T, B).contiguous(),
           "traj_index": torch.zeros(T, B, dtype=torch.int64), "an_object_is_in_hand": torch.zeros(T, B, 1, dtype=torch.int64)}
    batch = {"actions": torch.randint(0, 20, (T, B)), "old_action_log_probs": torch.full((T, B), -3.0), "adv_targ": torch.randn(T, B, 1),
             "c_adv_targ": torch.randn(T, B, 1), "returns": torch.randn(T, B, 1), "values": torch.randn(T, B, 1), "c_returns": torch.randn(T, B, 1)}
    pa, mk = torch.randint(0, 20, (T, B)), torch.ones(T, B, 1)
    t0 = time.time()
    opt.zero_grad()
    out, _ = m(obs, None, pa, mk)
    total, _ = ref_loss.safe_ppo_log_grad(out["logits"], out["values"], batch, 0.1)
    (total + ref_loss.safe_ppo_value(out["c_values"], batch["c_returns"])).backward()
    torch.nn.utils.clip_grad_norm_(params, 0.5)
    opt.step()
    dt = time.time() - t0
    return {"value": T * B / (4.0 * dt), "unit": "env-steps/s", "cores": n, "kind": "port",
            "sample": f"fp32 torch-CPU oracle, 1 of 4 epochs timed on T={T} x B={B} rows ({dt:.1f} s), L={L}, " + ("train mode (dropout 0.1)" if train_mode else "eval mode") + ", scaled x4"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--T", type=int, default=256)
    ap.add_argument("--envs-per-gpu", type=int, default=32)
    ap.add_argument("--L", type=int, default=12)
    ap.add_argument("--task", default="Fetch")
    ap.add_argument("--env-chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off (the reference trains with the policy in train() mode: default here too)")
    args = ap.parse_args()

    from safevla_amd import ops, parallel
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    rank, local, world = parallel.init_from_env()
    assert world == max(1, args.gpus) or world == 1, (world, args.gpus)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(1234 + rank)
    model = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
    model.train(not args.eval_mode)
    if world > 1:  # identical initial weights on every rank
        torch.distributed.broadcast(model.arena.flat_p, src=0)
        for t in model.towers:
            for p in t.visual_encoder.text_encoder.parameters():
                torch.distributed.broadcast(p.data, src=0)
        model.sync_weights()
    T, B = args.T, args.envs_per_gpu
    cfg = PPOLagConfig(env_chunk=args.env_chunk or None)
    eng = PPOLagEngine(model, cfg)
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=args.L, task=args.task, seed=1234 + rank), device=dev)

    def step():
        return eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])

    for _ in range(args.warmup):
        info = step()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = step()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    ms = dt / args.steps * 1e3
    env_steps = T * B * world
    S, R = 169 + args.L, T * B
    U = int(st.observations["goal_token_ids"][:T].reshape(R, -1).unique(dim=0).shape[0])
    algo = flops_per_update(R, S, args.L, U, cfg.update_repeats)

    roof = None
    gt = None
    if not args.no_roofline:
        # one extra, instrumented update: EVERY rank runs it (it contains the gradient / cost all-reduces), rank 0 times its
        # MFMA-kernel launches with HIP events
        if rank == 0:
            with GemmTimer(ops) as gt:
                step()
        else:
            step()
        parallel.barrier()
    if rank == 0 and gt is not None:
        allk = gt.summary()
        g = allk["gemm_nt256"]
        executed = sum(v["flops"] for v in allk.values())
        traffic = None   # HBM bytes per launch from the rocprofv3 PMC passes of this same command (profiles/, collected offline)
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01g_pmc_hbm_traffic.json")))["kernels"]
            inst = [v for k, v in pm.items() if "gemm_nt256" in k]      # one entry per epilogue instantiation: launch-weighted mean
            traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in inst) / sum(v["launches"] for v in inst))
        except Exception:
            pass
        roof = {"bound": "mfma", "kernel": "gemm_nt256k64_bf16_kernel (svla_gemm_nt_bf16, persistent 256x256 tile, BK=64)", "achieved": round(g["tflops"], 1),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(g["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, profiles/r01g_pmc_hbm_traffic.json; launch-weighted over the epilogue instantiations)",
                "launches_per_update": g["launches"], "avg_launch_ms": round(g["avg_ms"], 4), "flops_per_launch": g["flops"] / max(1, g["launches"]),
                "share_of_update": round(g["total_s"] / (ms * 1e-3), 3),
                "other_mfma_kernels": {k: {"achieved_tflops": round(v["tflops"], 1), "launches": v["launches"], "avg_launch_ms": round(v["avg_ms"], 4),
                                           "share_of_update": round(v["total_s"] / (ms * 1e-3), 3)} for k, v in allk.items() if k != "gemm_nt256"},
                "executed_mfma_tflop_per_update": round(executed / 1e12, 1),
                "executed_mfma_tflops_sustained": round(executed / (ms * 1e-3) / 1e12, 1)}
    cpu = None
    acting = None
    ns = None
    if rank == 0 and world == 1 and not args.no_roofline:
        acting = acting_bench(model, st, B, dev)
        ns = north_star_probe(model, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(L=args.L, train_mode=not args.eval_mode)
        one = cpu_baseline(T=8, B=2, L=args.L, train_mode=not args.eval_mode, threads=1)     # also at n = 1
        cpu["single_thread"] = {"value": one["value"], "unit": one["unit"], "sample": one["sample"]}
    if rank == 0:
        out = {"metric": "env-steps/sec through PPO-Lagrangian update", "value": round(env_steps / (ms * 1e-3), 1), "unit": "env-steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"C4-shard: {args.task}, T={T}-step rollout x {B} envs/GPU (BASELINE configs[3] per-GPU shard), "
                                      f"L={args.L} goal tokens, S={S} fusion tokens, 3 towers x 4 epochs x 1 minibatch, Adam+clip",
                          "global_envs": B * world, "rollout_steps": T, "rows_per_gpu": R, "parallelism": f"dp{world}",
                          "stage_losses": list(cfg.stage_losses), "weights": "random-init, reference geometry (168.9 M params)",
                          "dropout": 0.0 if args.eval_mode else 0.1},
               "reference_equivalent_tflop_per_update": round(algo / 1e12, 1),
               "note": "reference_equivalent counts SURVEY 8(d) FLOPs of the reference's schedule; the engine executes fewer (last fusion "
                       "layer only for the consumed token, T5 once per unique goal) -- see roofline.executed_mfma_*",
               "loss": {k: (round(v, 5) if isinstance(v, float) else v) for k, v in info.items()},
               "roofline": roof, "cpu_baseline": cpu, "acting": acting, "north_star_batch256": ns}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
