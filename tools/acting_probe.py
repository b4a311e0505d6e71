#!/usr/bin/env python3
"""Only the acting-path secondary measurement of bench.py (frozen DINOv2 ViT-S/14 on 2 frames per env step + one KV-cached 3-tower step,
64 envs), for rocprofv3 runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout
dev = torch.device("cuda")
torch.manual_seed(1234)
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
B = 64
st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=32, B=B, L=12, task="PickUp", seed=1234), device=dev)
print(json.dumps(bench.acting_bench(m, st, B, dev)))
