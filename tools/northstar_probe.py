#!/usr/bin/env python3
"""Only the north-star secondary probe of bench.py (3-tower fwd+bwd at 256 rows x 233 tokens), for rocprofv3 runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
dev = torch.device("cuda")
torch.manual_seed(1234)
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
print(json.dumps(bench.north_star_probe(m, dev)))
