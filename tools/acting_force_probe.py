import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout
from safevla_amd._lib import lib
dev = torch.device("cuda")
torch.manual_seed(1234)
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
B = 64
st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=32, B=B, L=12, task="PickUp", seed=1234), device=dev)
r = bench.acting_bench(m, st, B, dev)
print("SVLA_NT256_MIN_TILES", os.environ.get("SVLA_NT256_MIN_TILES", "default (160)"), {k: round(v) for k, v in r.items() if isinstance(v, float)}, flush=True)
