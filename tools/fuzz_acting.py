#!/usr/bin/env python3
"""Randomised acting rollouts across the KV-cache window boundary: models built with a SMALL cache window (``max_steps`` = 5 / 8 / 17 instead of 500) are stepped for up to
three windows' worth of single-step forwards over synthetic rollouts with episode boundaries (random envs B, goal tokens L, task sampler) three ways -- the recorded
launch plans (the default acting path: step counter, KV slot and masks in device memory), the eagerly issued path, and the eager path of the fp32 verification mode -- and
the three must agree step for step: plans == eager to bf16 rounding of identical arithmetic, bf16 vs fp32 on the bf16 ladder.  The fixed tests step a dozen times inside
one 500-slot window; this walks the counter through the wrap (the step that fills the last slot, the restart of the window, episodes that straddle it).

    python tools/fuzz_acting.py [--seed 0] [--cases 9]
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

DEV = "cuda"


def steps(m, st, N, plans):
    for t in m.towers:
        t.time_step_counter, t._kv = 0, None
    m.time_step_counter = 0
    m.enable_acting_plans(plans)
    out = []
    with torch.no_grad():
        for t in range(N):
            o, _ = m({k: v[t:t + 1] for k, v in st.observations.items()}, None, st.prev_actions[t:t + 1], st.masks[t:t + 1])
            out.append((o.distributions.logits.float().cpu(), o.values.float().cpu(), o.c_values.float().cpu()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=9)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    bad = 0
    case = 0
    for W in (5, 8, 17):
        torch.manual_seed(args.seed)
        m16 = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, max_steps=W).eval()
        m32 = SafeDinoLLAMATxNavActorCriticSeparate(device=DEV, max_steps=W, precision="fp32").eval()
        m32.load_state_dict({k: v.detach().clone() for k, v in m16.state_dict().items()})
        for _ in range(max(1, args.cases // 3)):
            B = rng.choice([1, 2, 3, 5, 9, 16])
            L = rng.choice([1, 4, 12, 33, 64])
            N = rng.randint(W + 2, 3 * W + 2)
            task = rng.choice(["ObjectNav", "PickUp", "Fetch", "Mixed"])
            tag = f"window={W} B={B} L={L} steps={N} {task}"
            case += 1
            try:
                st, _, _ = fill_synthetic_rollout(m16, SynthSpec(T=N, B=B, L=L, task=task, seed=500 + case, max_steps=rng.choice([max(3, W - 1), 2 * W])), device=DEV)      # episodes shorter than the window, or outliving it
                plan, eager, ref = steps(m16, st, N, True), steps(m16, st, N, False), steps(m32, st, N, False)
                e1 = e2 = 0.0
                for t in range(N):
                    for a, b, c in zip(plan[t], eager[t], ref[t]):
                        sc = max(1.0, float(c.abs().max()))
                        e1 = max(e1, float((a - b).abs().max()) / sc)
                        e2 = max(e2, float((a - c).abs().max()) / sc)
                fin = all(bool(torch.isfinite(x).all()) for s in plan for x in s)
                ok = fin and e1 < 2e-2 and e2 < 3e-2
                print(f"{'ok  ' if ok else 'FAIL'} {tag}: plans vs eager {e1:.2e}, bf16 vs fp32 mode {e2:.2e} (relative to max, worst step), finite={fin}", flush=True)
                bad += 0 if ok else 1
            except Exception as e:
                bad += 1
                print(f"FAIL {tag}: raised {e!r}"[:400], flush=True)
        del m16, m32
        torch.cuda.empty_cache()
    print(f"{bad} failing rollout(s) of {case} (seed {args.seed})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
