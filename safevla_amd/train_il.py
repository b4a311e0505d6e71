"""``python -m safevla_amd.train_il`` -- imitation-learning training loop on synthetic trajectories with the reference's flags
(/root/reference/training/offline/train_pl.py:24-71: --model_version --lr --per_gpu_batch --sliding_window --max_samples --input_sensors
--save_every --output_dir).  The reference's HDF5 + mp4 dataset (training/offline/chores_dataset.py) needs assets that are not available
offline; batches here have the same keys / shapes / vocabularies (pre-encoded DINOv2 features by default, ``--raw_frames`` for uint8 frames
through the frozen ViT), episodes of random length padded to the sliding window exactly as the dataset pads them (last_actions = 21,
actions = -1).  One process per GPU; gradients all-reduced over RCCL like the RL engine."""
import argparse
import json
import os
import time

import torch

from . import ops, parallel
from .il import MANIP, NAV, PAD_TOKEN, START_TOKEN, EarlyFusionCnnTransformer, ILTrainer


def synthetic_batch(B: int, T: int, L: int, device, generator: torch.Generator, raw_frames: bool = False, feat_dim: int = 384, siglip: bool = False):
    """``siglip``: 256 x 256 frames and the open_clip tokenizer's goal format (an id tensor [B, 64] padded with 1, preprocessors.py:300-343)."""
    g = generator
    r = lambda *s, hi: torch.randint(0, hi, s, device=device, generator=g)
    if raw_frames:
        H, W = (256, 256) if siglip else (224, 384)
        nav, man = r(B, T, H, W, 3, hi=256).to(torch.uint8), r(B, T, H, W, 3, hi=256).to(torch.uint8)
    else:
        nav, man = (torch.randn(B, T, feat_dim, 7, 12, device=device, generator=g) for _ in range(2))
    valid = torch.randint(max(2, T // 2), T + 1, (B,), device=device, generator=g)
    tt = torch.arange(T, device=device)[None, :].expand(B, T)
    pad = tt >= valid[:, None]
    actions = r(B, T, hi=20)
    last = torch.cat([torch.full((B, 1), START_TOKEN, device=device), actions[:, :-1]], dim=1)
    last = torch.where(pad, torch.full_like(last, PAD_TOKEN), last)
    actions = torch.where(pad, torch.full_like(actions, -1), actions)
    n_tok = torch.randint(3, L + 1, (B,), device=device, generator=g)
    ids = torch.randint(3, 32000, (B, L), device=device, generator=g)
    lt = torch.arange(L, device=device)[None, :].expand(B, L)
    ids = torch.where(lt == (n_tok[:, None] - 1), torch.ones_like(ids), ids)
    am = (lt < n_tok[:, None]).to(torch.int64)
    ids = ids * am
    if siglip:
        ids64 = torch.ones(B, 64, device=device, dtype=ids.dtype)
        ids64[:, :L] = torch.where(am > 0, ids, torch.ones_like(ids))
    return {NAV: nav, MANIP: man, "time_ids": tt.contiguous(), "padding_mask": pad, "last_actions": last, "actions": actions,
            "an_object_is_in_hand": r(B, T, hi=3), "goals": ids64 if siglip else dict(input_ids=ids, attention_mask=am)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="EarlyFusionCnnTransformer")
    ap.add_argument("--model_version", default="small_3")
    ap.add_argument("--loss", default="action")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--per_gpu_batch", type=int, default=16)
    ap.add_argument("--sliding_window", type=int, default=50)
    ap.add_argument("--max_samples", type=int, default=512)
    ap.add_argument("--save_every", type=int, default=0)
    ap.add_argument("--output_dir", default="gpurun_out/il")
    ap.add_argument("--input_sensors", nargs="+", default=[NAV, MANIP, "last_actions", "an_object_is_in_hand"])
    ap.add_argument("--goal_tokens", type=int, default=12)
    ap.add_argument("--raw_frames", action="store_true")
    ap.add_argument("--init_ckpt", default=None)
    args = ap.parse_args()
    rank, local, world = parallel.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    model = EarlyFusionCnnTransformer.build_model(args.model_version, args.input_sensors, args.loss, device=dev, ckpt_pth=args.init_ckpt)
    if world > 1:
        torch.distributed.broadcast(model.arena.flat_p, src=0)
        model.sync_weights()
    tr = ILTrainer(model, lr=args.lr)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    B, T = args.per_gpu_batch, args.sliding_window
    steps = max(1, args.max_samples // (B * world))
    t0 = time.perf_counter()
    for it in range(steps):
        batch = synthetic_batch(B, T, args.goal_tokens, dev, gen, args.raw_frames, feat_dim=model.dino_dim, siglip=model.text_encoder_name.startswith("SigLIP"))
        model.zero_grad()
        out = model(batch)
        (out["loss"] / world).backward()
        parallel.allreduce_sum_(model.arena.flat_g)
        tr.step_count += 1
        ar = model.arena
        ops.adam_step(ar.flat_p, ar.flat_g, ar.flat_m, ar.flat_v, ar.flat_bf16, tr.lr, tr.step_count, weight_decay=tr.wd)
        model.refresh_transposes()
        if rank == 0 and (it % max(1, steps // 8) == 0 or it == steps - 1):
            print(json.dumps({"step": it + 1, "loss": round(float(out["loss"]), 5)}), flush=True)
        if rank == 0 and args.save_every and (it + 1) % args.save_every == 0:
            os.makedirs(args.output_dir, exist_ok=True)
            torch.save({"state_dict": {"model." + k: v.detach().cpu() for k, v in model.state_dict().items()}, "global_step": it + 1},
                       os.path.join(args.output_dir, f"step_{it + 1}.ckpt"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"steps": steps, "timesteps_per_s": round(steps * B * T * world / dt, 1), "trajectories_per_s": round(steps * B * world / dt, 2),
                          "per_gpu_batch": B, "sliding_window": T, "n_gpus": world}), flush=True)


if __name__ == "__main__":
    main()
