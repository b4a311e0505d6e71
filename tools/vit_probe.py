#!/usr/bin/env python3
"""Only the frozen DINOv2 ViT-S/14 preprocessor on 128 uint8 frames (2 cameras x 64 envs), a few passes (for rocprofv3 runs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd.preproc import DataAugmentationPreprocessor, DinoViTPreprocessor
dev = torch.device("cuda")
torch.manual_seed(0)
vit = DinoViTPreprocessor("rgb_raw", "rgb_dinov2", device=dev)
fr = torch.randint(0, 256, (128, 224, 384, 3), device=dev, dtype=torch.uint8)
vit.process({"rgb_raw": fr}); torch.cuda.synchronize()
aug = DataAugmentationPreprocessor("rgb_raw", "rgb_norm", device=dev)      # the stand-alone u8 -> normalised fp32 frame path (dino_preprocessors.py:224-239)
for _ in range(4): aug.process({"rgb_raw": fr})
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("VIT_PASSES", 4))
for _ in range(n): vit.process({"rgb_raw": fr})
torch.cuda.synchronize()
print(f"{n * 128 / (time.perf_counter() - t0):.0f} frames/s")
