#!/usr/bin/env python3
"""Frozen DINOv2 ViT-S/14 preprocessor on 64 frames (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd.preproc import DinoViTPreprocessor
vit = DinoViTPreprocessor("rgb_raw", "rgb_dinov2", device="cuda")
fr = torch.randint(0, 256, (64, 224, 384, 3), device="cuda", dtype=torch.uint8)
for _ in range(5): vit.process({"rgb_raw": fr})
torch.cuda.synchronize()
