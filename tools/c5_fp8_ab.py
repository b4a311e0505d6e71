#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU's shard (32 envs x 256 steps, mixed tasks, 64-token instructions, S = 233): the update with the
fusion-encoder attention on the bf16 kernels vs on the fp8 MFMA kernels -- time per update, and the agreement of the gradients."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.engine import PPOLagConfig, PPOLagEngine
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout
dev = torch.device("cuda")
torch.manual_seed(1234)
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
res = {}
for fp8 in (False, True):
    res["fp8" if fp8 else "bf16"] = bench.secondary_config(m, dev, "C5-shard", "Mixed", 256, 32, 64, None, 2.31964, fp8)
m.eval()
T, B = 256, 32
st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=T, B=B, L=64, task="Mixed", seed=5), device=dev)
st.compute_returns(nxt["next_value"], nxt["next_c_value"])
g = {}
for fp8 in (False, True):
    m.set_fp8_attention(fp8)
    eng = PPOLagEngine(m, PPOLagConfig(env_chunk=None, cost_limit=2.31964))
    m.zero_grad(); eng._sums.zero_()
    eng._accumulate(st.batch_slice(0, B), T * B, 0.2, last=True)
    g[fp8] = m.arena.flat_g.double().clone()
    del eng
m.set_fp8_attention(False)
a, b = g[False], g[True]
res["flat_gradient_fp8_vs_bf16"] = {"cosine": round(torch.nn.functional.cosine_similarity(a, b, dim=0).item(), 5), "rel_l2": round(((a - b).norm() / a.norm()).item(), 4), "mode": "eval (dropout off), one pass over the minibatch"}
print(json.dumps(res))
