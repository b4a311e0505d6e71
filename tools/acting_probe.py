#!/usr/bin/env python3
"""Where does a single acting step (B envs, KV-cached 3-tower forward) spend its wall time?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate, _TowerFn
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout
dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=40, B=B, L=12, task="Fetch", seed=1), device=dev)
for t in m.towers: t.time_step_counter, t._kv = 0, None
step_in = lambda t: ({k: v[t:t + 1] for k, v in st.observations.items()}, st.prev_actions[t:t + 1], st.masks[t:t + 1])
tp = tf = 0.0
with torch.no_grad():
    for t in range(36):
        o, pa, mk = step_in(t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prep = m.prepare(o, pa, mk)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for tw in m.towers: tw.run_forward(prep, need_grad=False)
        t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        if t >= 4: tp += t1 - t0; tf += t3 - t1; cpu_f = t2 - t1
print(f"B={B}: prepare {tp/32*1e3:.3f} ms/step, 3 towers {tf/32*1e3:.3f} ms/step (CPU issue time of the last step {cpu_f*1e3:.3f} ms)")
