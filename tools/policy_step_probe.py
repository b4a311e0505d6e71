#!/usr/bin/env python3
"""Only the KV-cached single-step 3-tower forward of the acting path (recorded launch plans, three streams), N steps at 64 envs -- for rocprofv3 --kernel-trace --stats:
kernel time per step by kernel.  python tools/policy_step_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout
dev = torch.device("cuda")
torch.manual_seed(1234)
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
B, n = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 40
st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=n + 8, B=B, L=12, task="PickUp", seed=1234), device=dev)
step_in = lambda t: ({k: v[t:t + 1] for k, v in st.observations.items()}, None, st.prev_actions[t:t + 1], st.masks[t:t + 1])
for t in m.towers:
    t.time_step_counter, t._kv = 0, None
m.enable_acting_plans(True)
with torch.no_grad():
    for t in range(4):
        m(*step_in(t))
    torch.cuda.synchronize()
    print("MARK steady state starts", flush=True)
    t0 = time.perf_counter()
    for t in range(4, 4 + n):
        m(*step_in(t))
    torch.cuda.synchronize()
print(f"{n * B / (time.perf_counter() - t0):.0f} env-steps/s, {1e3 * (time.perf_counter() - t0) / n:.3f} ms per step")
