"""``train`` entry point with the reference's command-line flags
(/root/reference/scripts/train.sh:116-136 -> training/online/dinov2_vits_tsfm_base.py:395-402 ``fire.Fire(...)`` ->
OnPolicyRunnerMixin.train, training/online/allenact_trainer.py:47-72).  Flags mirror the dataclass fields of
``BaseConfigParams`` (training/online/base.py:123-132), ``DinoV2ViTSTSFMBaseParams`` (dinov2_vits_tsfm_base.py:60-88) and
``OnPolicyRunnerMixin`` (allenact_trainer.py:11-23).  AI2-THOR is replaced by the synthetic generator (simulator off the
critical path); everything else -- stage schedule, losses, Adam/clip, lambda update, checkpoints -- follows the reference
pipeline (dinov2_vits_tsfm_base.py:293-380).

    python -m safevla_amd.train train --num_train_processes 32 --cost_limit 2.31964 --output_dir out --tag run [--il_ckpt_path ..]
    torchrun --nproc-per-node 8 -m safevla_amd.train train --num_train_processes 256 ...
"""
import argparse
import json
import os
import time

import torch


def build_parser():
    ap = argparse.ArgumentParser(prog="safevla_amd.train")
    ap.add_argument("mode", choices=["train"])
    # BaseConfigParams / DinoV2ViTSTSFMBaseParams
    ap.add_argument("--num_train_processes", type=int, default=32)
    ap.add_argument("--distributed_nodes", type=int, default=1)
    ap.add_argument("--dataset_dir", default="data/fifteen/ObjectNavType")
    ap.add_argument("--max_steps", type=int, default=500)
    ap.add_argument("--tag", default="SafeVLA-ObjectNavType-RL-DinoV2-ViTS-TSFM")
    ap.add_argument("--lr", type=float, default=2e-5)
    ap.add_argument("--save_interval", type=int, default=50_000)
    ap.add_argument("--il_ckpt_path", default=None)
    ap.add_argument("--wandb_project", default="")
    ap.add_argument("--wandb_entity", default="")
    # OnPolicyRunnerMixin
    ap.add_argument("--output_dir", default="/root/results")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--machine_id", type=int, default=0)
    ap.add_argument("--distributed_ip_and_port", default="127.0.0.1:0")
    ap.add_argument("--callbacks", default="")
    ap.add_argument("--save_dir_fmt", default="flat", choices=["flat", "nested"])
    ap.add_argument("--extra_tag", default="")
    ap.add_argument("--deterministic_cudnn", default=False)
    ap.add_argument("--deterministic_agents", default=False)
    ap.add_argument("--disable_tensorboard", default=True)
    ap.add_argument("--disable_config_saving", default=True)
    ap.add_argument("--restart_pipeline", default=False)
    ap.add_argument("--cost_limit", type=float, default=None)
    ap.add_argument("--checkpoint", default=None)
    # synthetic-run controls (not in the reference)
    ap.add_argument("--num_steps", type=int, default=128, help="rollout length (reference: TrainingSettings(num_steps=128))")
    ap.add_argument("--total_steps", type=int, default=0, help="stop after this many env steps (0: run the full stage schedule)")
    ap.add_argument("--task", default=None, help="ObjectNav | PickUp | Fetch | Mixed (env e -> task e mod 3); default: inferred from --tag / --dataset_dir")
    ap.add_argument("--collect", default="acting", choices=["acting", "teacher"],
                    help="acting: step the synthetic vector env through the KV-cached single-step policy like the reference engine; "
                         "teacher: fill the storage with one full-sequence pass (benchmark mode)")
    ap.add_argument("--goal_tokens", type=int, default=12)
    return ap


def infer_task(tag: str, dataset_dir: str) -> str:
    """The reference picks its task sampler from the dataset directory (``.../ObjectNavType`` | ``PickupType`` | ``FetchType``,
    scripts/train.sh:97-110; tag = the same string)."""
    s = f"{tag} {dataset_dir}".lower()
    for key, task in (("pickup", "PickUp"), ("fetch", "Fetch"), ("objectnav", "ObjectNav"), ("mixed", "Mixed")):
        if key in s:
            return task
    return "ObjectNav"


def stage_for(step: int):
    """pipeline_stages of dinov2_vits_tsfm_base.py:348-379: 200k critics-only, then ppo_log_loss."""
    if step < 200_000:
        return ("ppo_value_loss", "safe_ppo_value_loss")
    return ("ppo_log_loss",)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.task is None:
        args.task = infer_task(args.tag, args.dataset_dir)
    from . import parallel
    from .checkpoint import init_towers_from_il, load_checkpoint, save_checkpoint
    from .engine import PPOLagConfig, PPOLagEngine
    from .model import SafeDinoLLAMATxNavActorCriticSeparate
    from .storage import RolloutStorage
    from .synth_env import SynthSpec, SynthVectorEnv, collect_rollout, fill_synthetic_rollout

    rank, local, world = parallel.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # rank-independent seed while the model is built: the frozen T5 encoder and the arena are drawn from the global RNG and every
    # replica must hold the same function; per-rank streams (sampling, synthetic environments) start afterwards
    torch.manual_seed(args.seed or 0)
    env0, B = parallel.shard_envs(args.num_train_processes, world, rank)
    model = SafeDinoLLAMATxNavActorCriticSeparate(device=dev, max_steps=args.max_steps)
    torch.manual_seed((args.seed or 0) + 1 + rank)
    if args.il_ckpt_path:
        init_towers_from_il(model, args.il_ckpt_path)
    cfg = PPOLagConfig(lr=args.lr, cost_limit=args.cost_limit if args.cost_limit is not None else 1e9)
    eng = PPOLagEngine(model, cfg)
    step = 0
    if args.checkpoint:
        ck = torch.load(args.checkpoint, map_location="cpu")
        load_checkpoint(ck, model, eng)
        step = int(ck.get("total_steps", 0))
    parallel.broadcast_model_(model)      # belt and braces: every parameter and buffer (incl. the frozen text encoder) from rank 0
    os.makedirs(args.output_dir, exist_ok=True)
    budget = args.total_steps or int(1e9)
    next_save = step + args.save_interval
    T = args.num_steps
    env = st = None
    if args.collect == "acting":
        env = SynthVectorEnv(B, L=args.goal_tokens, task=args.task, seed=1234 + rank, max_steps=args.max_steps, device=dev, env_offset=env0)
        st = RolloutStorage(T, device=dev, store_tokens=True)
        st.initialize(env.reset(), num_samplers=B)
    while step < budget:
        t0 = time.time()
        if env is not None:
            nxt = collect_rollout(model, env, st, T)
            s_, n_ = env.pop_episode_costs()
            ep = dict(episode_cost_sum=s_, n_episodes=n_)
        else:
            st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=args.goal_tokens, task=args.task, seed=1234 + rank + step,
                                                                  max_steps=args.max_steps, env_offset=env0), device=dev)
        cfg.stage_losses = stage_for(step)
        info = eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])
        if env is not None:
            st.after_updates()
        step += info["env_steps"]
        if rank == 0:
            info.update(training_step=step, env_steps_per_s=info["env_steps"] / (time.time() - t0), stage=list(cfg.stage_losses))
            print(json.dumps({k: (round(v, 5) if isinstance(v, float) else v) for k, v in info.items()}), flush=True)
            if step >= next_save or step >= budget:
                save_checkpoint(os.path.join(args.output_dir, f"exp_{args.tag}__stage_{0 if step < 200_000 else 1}__steps_{step:012d}.pt"), model, eng, step)
                next_save += args.save_interval
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
