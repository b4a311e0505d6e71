#!/usr/bin/env python3
"""Only the collect-then-update secondary of bench.py (the whole training iteration through the acting path), + a split of one collected step by stage (synchronised: indicative)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.storage import RolloutStorage
from safevla_amd.synth_env import SynthVectorEnv
dev = torch.device("cuda")
torch.manual_seed(1234)
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
print(json.dumps(bench.collect_then_update(m, dev)))
B, T = 64, 64
env = SynthVectorEnv(B, L=12, task="PickUp", seed=99, max_steps=500, device=dev)
st = RolloutStorage(T, device=dev, store_tokens=True)
st.initialize(env.reset(), num_samplers=B)
acc = dict(inp=0.0, model=0.0, sample=0.0, env=0.0, add=0.0)
def tick():
    torch.cuda.synchronize(); return time.perf_counter()
with torch.no_grad():
    for i in range(T):
        t0 = tick(); inp = st.agent_input_for_next_step()
        t1 = tick(); out, _ = m(inp["observations"], None, inp["prev_actions"], inp["masks"])
        t2 = tick(); actions = out.distributions.sample().reshape(B); logp = out.distributions.log_prob(actions.reshape(1, B)).reshape(B)
        t3 = tick(); obs, reward, cost, done, _ = env.step(actions)
        t4 = tick(); st.add(obs, None, actions, logp, out.values.reshape(B, 1), reward.reshape(B, 1), cost.reshape(B, 1), out.c_values.reshape(B, 1), (1.0 - done.float()).reshape(B, 1))
        t5 = tick()
        if i >= 8:
            for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): acc[k] += v
print({k: round(v / (T - 8) * 1e3, 3) for k, v in acc.items()}, "ms per step (synchronised after every stage)")
