"""TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the frozen ViT preprocessor paths (DINOv2 ViT-S/B/L-14, SigLIP ViT-B/16).

Reference call chain: DataAugmentationPreprocessor.process (/root/reference/architecture/allenact_preprocessors/
dino_preprocessors.py:224-239) -> DinoViTEmbedder.forward (:27-35).  The ViT itself is third-party
(``torch.hub.load("facebookresearch/dinov2", "dinov2_vits14")``, dino_preprocessors.py:106; not vendored, no network):
its published forward_features is restated here -- **parity unpinned** against DINOv2 proper.  Same for the SigLIP twin
(architecture/allenact_preprocessors/siglip_preprocessors.py:18-104: open_clip ``hf-hub:timm/ViT-B-16-SigLIP-256`` -> ``visual.trunk``
= a timm VisionTransformer without class token / LayerScale; 256 x 256 input, mean = std = 0.5).
"""
import torch
import torch.nn.functional as F

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def normalize(frames_u8: torch.Tensor) -> torch.Tensor:
    x = frames_u8.permute(0, 3, 1, 2).float() / 255.0
    x = (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)
    return x.permute(0, 2, 3, 1)


def vit_features(sd, frames_u8: torch.Tensor, heads=6, native_grid=37, patch=14, crop_x=3, mean=MEAN, std=STD):
    """sd: state_dict with DINOv2 / timm names (fp32 CPU).  Returns (normed tokens [B, (1+)gh*gw, dim], pooled [B, dim, 7, 12]).
    Class token, LayerScale and position-embedding interpolation are applied when the state dict has them (DINOv2) and skipped
    when it does not (timm SigLIP trunk: ``forward_features`` = patch_embed + pos_embed -> blocks -> norm)."""
    x = frames_u8.permute(0, 3, 1, 2).float() / 255.0
    x = (x - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    if crop_x:
        x = x[:, :, :, crop_x:-crop_x]
    B = x.shape[0]
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    gh, gw = x.shape[-2:]
    dim = x.shape[1]
    x = x.flatten(2).transpose(1, 2)
    pe = sd["pos_embed"][0]
    nc = 1 if "cls_token" in sd else 0
    if (gh, gw) != (native_grid, native_grid):
        pp = pe[nc:].reshape(1, native_grid, native_grid, dim).permute(0, 3, 1, 2)
        pp = F.interpolate(pp, size=(gh, gw), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).reshape(gh * gw, dim)
        pe = torch.cat([pe[:nc], pp], 0)
    if nc:
        x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], 1)
    x = x + pe[None]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    hd = dim // heads
    one = torch.ones(dim)
    for i in range(depth):
        p = lambda n: sd[f"blocks.{i}.{n}"]
        g1, g2 = (p("ls1.gamma"), p("ls2.gamma")) if f"blocks.{i}.ls1.gamma" in sd else (one, one)
        h = F.layer_norm(x, (dim,), p("norm1.weight"), p("norm1.bias"), 1e-6)
        qkv = F.linear(h, p("attn.qkv.weight"), p("attn.qkv.bias")).view(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        a = torch.softmax(qkv[0] @ qkv[1].transpose(-1, -2) / hd ** 0.5, -1) @ qkv[2]
        x = x + g1 * F.linear(a.transpose(1, 2).reshape(B, -1, dim), p("attn.proj.weight"), p("attn.proj.bias"))
        h = F.layer_norm(x, (dim,), p("norm2.weight"), p("norm2.bias"), 1e-6)
        x = x + g2 * F.linear(F.gelu(F.linear(h, p("mlp.fc1.weight"), p("mlp.fc1.bias"))), p("mlp.fc2.weight"), p("mlp.fc2.bias"))
    x = F.layer_norm(x, (dim,), sd["norm.weight"], sd["norm.bias"], 1e-6)
    pooled = F.adaptive_avg_pool2d(x[:, nc:].permute(0, 2, 1).reshape(B, dim, gh, gw), (7, 12))
    return x, pooled
