#!/usr/bin/env python3
"""Randomised configurations through the whole three-tower path: rollouts of random shape (T steps x B envs x L goal tokens, any task sampler, random env-chunking)
through ``PPOLagEngine._accumulate`` on the bf16 product path AND on the fp32 verification mode (the mode the reference goldens pin at 1e-6) on the same weights and
rollout -- loss sums and the 62.9 M-element gradient must agree on the bf16 ladder -- followed by one complete ``update`` (GAE, lambda, epochs x minibatches, clip + Adam;
small minibatches are recorded in the first epoch and replayed in the others) whose parameters must stay finite.  The fixed tests run the BASELINE configurations; this looks
for the (T, B, L, chunk) combination nobody listed: T = 1, B = 1, one goal token, 64 goal tokens (S = 233: other attention tiles), ragged last env-chunk, more minibatches
than chunks.  One line per configuration, exit code 1 on any violation.

    python tools/fuzz_engine.py [--seed 0] [--cases 24] [--critic-type linear|discrete|mlp]
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from safevla_amd.engine import PPOLagConfig, PPOLagEngine
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

DEV = "cuda"


def accumulate(model, st, T, B, chunk, lam):
    eng = PPOLagEngine(model, PPOLagConfig(env_chunk=chunk))
    model.zero_grad()
    eng._sums.zero_()
    c = chunk or B
    for c0 in range(0, B, c):
        eng._accumulate(st.batch_slice(c0, min(B, c0 + c)), T * B, lam, last=c0 + c >= B)
    torch.cuda.synchronize()
    return model.arena.flat_g.double().clone(), eng._sums.clone()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--critic-type", default="linear", choices=["linear", "discrete", "mlp"], help="critic heads of both value towers (allenact_dino_transformer.py:147-162,720-766)")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    torch.manual_seed(args.seed)
    m16 = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, critic_type=args.critic_type).eval()
    m32 = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, precision="fp32", critic_type=args.critic_type).eval()
    sd0 = {k: v.detach().clone() for k, v in m16.state_dict().items()}
    m32.load_state_dict(sd0)
    bad = 0
    for i in range(args.cases):
        T = rng.choice([1, 2, 3, 5, 8, 13, 16, 33, 64])
        B = rng.choice([1, 2, 3, 4, 5, 8, 13])
        L = rng.choice([1, 2, 4, 7, 12, 20, 33, 64])
        task = rng.choice(["ObjectNav", "PickUp", "Fetch", "Mixed"])
        chunk = rng.choice([None, None, 1, 2, 3, 5])
        chunk = None if (chunk is not None and chunk >= B) else chunk
        lam = rng.choice([0.0, 0.3, 2.0])
        nmb = rng.choice([1, 1, 2, 3])
        if i < 2:       # always in the campaign: a ONE-step rollout (the decoder's T = 1 case must not take the KV-cached acting branch when gradients are needed)
            T, B, chunk = 1, (3, 1)[i], None
        tag = f"T={T} B={B} L={L} {task} chunk={chunk} lam={lam} minibatches={nmb}"
        try:
            m16.load_state_dict(sd0)
            m16.arena.flat_m.zero_()
            m16.arena.flat_v.zero_()
            st, nxt, ep = fill_synthetic_rollout(m16, SynthSpec(T=T, B=B, L=L, task=task, seed=100 + i), device=DEV)
            st.compute_returns(nxt["next_value"], nxt["next_c_value"])
            g16, s16 = accumulate(m16, st, T, B, chunk, lam)
            g32, s32 = accumulate(m32, st, T, B, chunk, lam)
            cos = torch.nn.functional.cosine_similarity(g16, g32, dim=0).item()
            rel = ((g16 - g32).norm() / g32.norm()).item()
            ds = np.abs(s16.cpu().numpy()[[0, 1, 2, 4]] - s32.cpu().numpy()[[0, 1, 2, 4]])
            tol = 1e-2 * np.abs(s32.cpu().numpy()[[0, 1, 2, 4]]) + 2e-3 * T * B + 2e-2 * np.sqrt(T * B)      # per-row bf16 error of v enters (v - R)^2 linearly: ~1e-2 per row, averaging as sqrt(rows)
            # a handful of rows: the value towers' gradient is a sum of few (v - R) terms, each carrying the bf16 forward's ~1e-2 relative error of v undamped by
            # averaging (per-tower relative error 0.1-0.25 at 2-25 rows -- more with the MLP critic head --, 5e-3 from ~100 rows up).  The actor tower has no such
            # cancellation (its gradient is advantage x d log pi): it is held to the tight bound at every size, the whole gradient from 64 rows up
            lo, hi = m16.arena.tower_ranges[0]
            cos_actor = torch.nn.functional.cosine_similarity(g16[lo:hi], g32[lo:hi], dim=0).item()
            ok = (cos_actor > 0.999 and (cos > 0.999 and rel < 5e-2 if T * B >= 64 else cos > 0.95)) and bool((ds <= tol).all()) and bool(torch.isfinite(g16).all())
            # one complete update on the product path: train mode (dropout), random minibatch count (recorded / replayed small minibatches)
            m16.train()
            # ... with the switches a deployment can flip, at random: fp8 MFMA attention (BASELINE configs[4]), the reference-faithful per-row T5 dropout, bitwise-repeatable accumulation
            fp8, per_row, det = rng.random() < 0.25, rng.random() < 0.3, rng.random() < 0.25
            m16.set_fp8_attention(fp8)
            m16.t5_dropout_per_row = per_row
            try:
                eng = PPOLagEngine(m16, PPOLagConfig(env_chunk=chunk, num_mini_batch=min(nmb, B), deterministic=det))
                info = eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])
            finally:
                m16.set_fp8_attention(False)
                m16.t5_dropout_per_row = False
                m16.eval()
            tag += f" | update: fp8={int(fp8)} t5_per_row={int(per_row)} deterministic={int(det)}"
            fin = bool(torch.isfinite(m16.arena.flat_p).all()) and all(np.isfinite(v) for v in info.values() if isinstance(v, float))
            ok = ok and fin and info["env_steps"] == T * B
            print(f"{'ok  ' if ok else 'FAIL'} {tag}: gradient cosine {cos:.6f} (actor tower {cos_actor:.6f}), rel L2 {rel:.2e}, loss-sum diffs {np.array2string(ds, precision=2)}; update finite={fin}, ppo_total {info['ppo_total']:.4f}", flush=True)
            bad += 0 if ok else 1
        except Exception as e:
            bad += 1
            print(f"FAIL {tag}: raised {e!r}"[:400], flush=True)
            m16.eval()
        del st, nxt
        torch.cuda.empty_cache()
    print(f"{bad} failing configuration(s) of {args.cases} (seed {args.seed}, critic_type {args.critic_type})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
