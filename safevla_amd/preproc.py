"""Frozen-ViT sensor preprocessors (rollout time), mirroring the reference's preprocessor contract
``process(obs: Dict[str, Tensor]) -> Tensor`` with attributes ``input_uuids`` / ``uuid``:

  * ``DataAugmentationPreprocessor``  /root/reference/architecture/allenact_preprocessors/dino_preprocessors.py:166-239
    (u8 HWC -> /255, -mean, /std; augmentation is a host-configured torchvision op list and is off for synthetic runs)
  * ``DinoViTPreprocessor`` / ``DinoViTEmbedder``  dino_preprocessors.py:20-125: crop W 384 -> 378, DINOv2 ViT-S/14
    ``forward_features(...)["x_norm_patchtokens"]`` -> (B,384,16,27) -> AdaptiveAvgPool2d((7,12)).

The DINOv2 network itself is third-party (``torch.hub facebookresearch/dinov2``, not in the reference tree, no network
here): its published ViT-S/14 forward is restated (pre-LN blocks with qkv/proj biases, LayerScale, GELU MLP, final LayerNorm,
bicubic position-embedding interpolation) with the hub model's ``state_dict`` names, random-init geometry. PARITY UNPINNED
against DINOv2 proper; pinned against the fp32 oracle restatement (oracle/ref_vit.py).

Geometry-generic: DINOv2 ViT-S/B/L-14 (widths 384 / 768 / 1024, 433 tokens) and the SigLIP ViT-B/L-16 trunk
(256 tokens, no class token) run on the same kernels -- ``SigLIPPreprocessor`` mirrors siglip_preprocessors.py:18-104.

MI355X path: normalise + crop + im2col fused in one kernel, patch embedding and all block linears on the bf16 MFMA GEMM
(LayerScale folded into the frozen weights at sync time), fused attention at S = 433, output written directly in the rollout
storage's bf16 token layout [B, ncam, 84, 384] (and/or the reference's fp32 (B,384,7,12)).
"""
import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .api import Box
from .model import _NS

DINO_RGB_MEANS = (0.48145466, 0.4578275, 0.40821073)
DINO_RGB_STDS = (0.26862954, 0.26130258, 0.27577711)
BF16 = torch.bfloat16


class DataAugmentationPreprocessor:
    def __init__(self, rgb_input_uuid: str, output_uuid: str, device="cuda", normalize=True, mean=DINO_RGB_MEANS, stdev=DINO_RGB_STDS,
                 height=224, width=384, use_augmentation=False, **kw):
        if use_augmentation:
            raise NotImplementedError("torchvision augmentation lists are simulator-side configuration (off for synthetic runs)")
        self.input_uuids, self.uuid, self.device = [rgb_input_uuid], output_uuid, torch.device(device)
        self.mean, self.stdev, self.normalize = mean, stdev, normalize
        self.observation_space = Box(-float("inf"), float("inf"), (height, width, 3))        # dino_preprocessors.py:205-214

    def to(self, device):
        self.device = torch.device(device)
        return self

    def process(self, obs: Dict[str, torch.Tensor], *a, **k) -> torch.Tensor:
        x = obs[self.input_uuids[0]].to(self.device)
        assert x.dtype == torch.uint8 and x.shape[-1] == 3
        return ops.normalize_u8(x.contiguous(), self.mean if self.normalize else (0, 0, 0), self.stdev if self.normalize else (1, 1, 1))


# geometry presets: (dim, depth, heads, patch, native_grid, class token, LayerScale)
VIT_PRESETS = {
    # DINOv2 (torch.hub facebookresearch/dinov2; dino_preprocessors.py:14-18,54-76): pos_embed is the 37 x 37 grid of 518 / 14
    "dinov2_vits14": dict(dim=384, depth=12, heads=6, patch=14, native_grid=37, cls=True, layerscale=True),
    "dinov2_vitb14": dict(dim=768, depth=12, heads=12, patch=14, native_grid=37, cls=True, layerscale=True),
    "dinov2_vitl14": dict(dim=1024, depth=24, heads=16, patch=14, native_grid=37, cls=True, layerscale=True),
    # timm trunk of hf-hub:timm/ViT-B-16-SigLIP-256 (siglip_preprocessors.py:15,86-88; image_encoders.py:75-112): 256 x 256 input,
    # 16 x 16 patches -> 256 tokens, no class token, no LayerScale, learned [1, 256, 768] position embedding used as is
    "ViT-B-16-SigLIP-256": dict(dim=768, depth=12, heads=12, patch=16, native_grid=16, cls=False, layerscale=False),
    "ViT-L-16-SigLIP-256": dict(dim=1024, depth=24, heads=16, patch=16, native_grid=16, cls=False, layerscale=False),
}


class DinoViT(nn.Module):
    """Frozen ViT trunk with the hub / timm parameter names: DINOv2 ViT-S/B/L-14 and the SigLIP ViT-B/L-16 geometries."""

    def __init__(self, device, dim=384, depth=12, heads=6, patch=14, native_grid=37, cls=True, layerscale=True):
        super().__init__()
        assert dim % 64 == 0 and dim // heads == 64, "attention kernels: head_dim 64"
        self.dim, self.depth, self.heads, self.patch, self.native_grid = dim, depth, heads, patch, native_grid
        self.has_cls, self.has_ls = cls, layerscale
        d = torch.device(device)
        P = lambda *s, sc=0.02: nn.Parameter((torch.randn(*s) * sc).to(d), requires_grad=False)
        if cls:
            self.cls_token = P(1, 1, dim)
            self.mask_token = P(1, dim)
        self.pos_embed = P(1, (1 if cls else 0) + native_grid * native_grid, dim)
        self.patch_embed = _NS(); self.patch_embed.proj = _NS()
        self.patch_embed.proj.weight = P(dim, 3, patch, patch, sc=1.0 / math.sqrt(3 * patch * patch))
        self.patch_embed.proj.bias = P(dim)
        self.blocks = nn.ModuleList()
        for _ in range(depth):
            b = _NS()
            b.norm1 = _NS(); b.norm1.weight = nn.Parameter(torch.ones(dim, device=d), requires_grad=False); b.norm1.bias = P(dim)
            b.attn = _NS(); b.attn.qkv = _NS(); b.attn.proj = _NS()
            b.attn.qkv.weight = P(3 * dim, dim, sc=1.0 / math.sqrt(dim)); b.attn.qkv.bias = P(3 * dim)
            b.attn.proj.weight = P(dim, dim, sc=1.0 / math.sqrt(dim)); b.attn.proj.bias = P(dim)
            if layerscale:
                b.ls1 = _NS(); b.ls1.gamma = nn.Parameter(torch.full((dim,), 1.0, device=d), requires_grad=False)
            b.norm2 = _NS(); b.norm2.weight = nn.Parameter(torch.ones(dim, device=d), requires_grad=False); b.norm2.bias = P(dim)
            b.mlp = _NS(); b.mlp.fc1 = _NS(); b.mlp.fc2 = _NS()
            b.mlp.fc1.weight = P(4 * dim, dim, sc=1.0 / math.sqrt(dim)); b.mlp.fc1.bias = P(4 * dim)
            b.mlp.fc2.weight = P(dim, 4 * dim, sc=1.0 / math.sqrt(4 * dim)); b.mlp.fc2.bias = P(dim)
            if layerscale:
                b.ls2 = _NS(); b.ls2.gamma = nn.Parameter(torch.full((dim,), 1.0, device=d), requires_grad=False)
            self.blocks.append(b)
        self.norm = _NS(); self.norm.weight = nn.Parameter(torch.ones(dim, device=d), requires_grad=False); self.norm.bias = P(dim)
        self._rt = None

    def interpolated_pos(self, gh: int, gw: int) -> torch.Tensor:
        """[(1 +) gh*gw, dim] fp32: class position + bicubic resize of the native_grid^2 patch positions (DINOv2 interpolate_pos_encoding);
        the native grid is used as is (SigLIP at its own 16 x 16)."""
        pe = self.pos_embed[0].float()
        g, nc = self.native_grid, (1 if self.has_cls else 0)
        if (gh, gw) == (g, g):
            return pe.contiguous()
        patch = pe[nc:].reshape(1, g, g, self.dim).permute(0, 3, 1, 2)
        patch = F.interpolate(patch, size=(gh, gw), mode="bicubic", align_corners=False)
        return torch.cat([pe[:nc], patch.permute(0, 2, 3, 1).reshape(gh * gw, self.dim)], 0).contiguous()

    def sync(self, gh=16, gw=27, KP=None):
        K = 3 * self.patch * self.patch
        KP = KP or ((K + 31) // 32) * 32          # im2col rows padded to the GEMM's K granule (588 -> 608; 768 stays)
        rt = dict(gh=gh, gw=gw, KP=KP)
        w = self.patch_embed.proj.weight.reshape(self.dim, -1).float()
        wp = torch.zeros(self.dim, KP, device=w.device)
        wp[:, : w.shape[1]] = w
        rt["pe_w"] = wp.to(BF16).contiguous()
        rt["pe_b"] = self.patch_embed.proj.bias.float().contiguous()
        rt["pos"] = self.interpolated_pos(gh, gw)
        rt["cls"] = self.cls_token.reshape(-1).float().contiguous() if self.has_cls else None
        blocks = []
        one = torch.ones(self.dim, device=w.device)
        for b in self.blocks:   # LayerScale folded into the frozen projections: gamma * (W x + b)
            g1, g2 = (b.ls1.gamma.float(), b.ls2.gamma.float()) if self.has_ls else (one, one)
            blocks.append(dict(qkv=b.attn.qkv.weight.to(BF16).contiguous(), qkv_b=b.attn.qkv.bias.float().contiguous(),
                               proj=(g1[:, None] * b.attn.proj.weight.float()).to(BF16).contiguous(), proj_b=(g1 * b.attn.proj.bias.float()).contiguous(),
                               fc1=b.mlp.fc1.weight.to(BF16).contiguous(), fc1_b=b.mlp.fc1.bias.float().contiguous(),
                               fc2=(g2[:, None] * b.mlp.fc2.weight.float()).to(BF16).contiguous(), fc2_b=(g2 * b.mlp.fc2.bias.float()).contiguous()))
        rt["blocks"] = blocks
        self._rt = rt

    @torch.no_grad()
    def patch_tokens(self, frames_u8: torch.Tensor, mean=DINO_RGB_MEANS, std=DINO_RGB_STDS, crop_x: int = 3) -> torch.Tensor:
        """frames_u8 [B,H,W,3] uint8 -> normed tokens [B, (1 +) gh*gw, dim] bf16 (class token first when the geometry has one).
        The patch grid is (H // patch) x ((W - 2 crop_x) // patch): 224 x 384 -> 16 x 27 (DINOv2), 256 x 256 -> 16 x 16 (SigLIP)."""
        B, H, W, _ = frames_u8.shape
        gh, gw = H // self.patch, (W - 2 * crop_x) // self.patch
        if self._rt is None or (self._rt["gh"], self._rt["gw"]) != (gh, gw):
            self.sync(gh, gw)
        rt = self._rt
        KP, C = rt["KP"], self.dim
        nc = 1 if self.has_cls else 0
        NP, S = gh * gw, gh * gw + nc
        dev = frames_u8.device
        cols = torch.empty(B * NP, KP, device=dev, dtype=BF16)
        ops.patchify_u8(frames_u8.contiguous(), mean, std, cols, crop_x=crop_x, P=self.patch, gh=gh, gw=gw)
        pt = ops.gemm_nt(cols, rt["pe_w"], B * NP, C, KP, bias=rt["pe_b"])
        # token rows padded to a whole number of 256-row GEMM panels (128 frames x 433 tokens = 216.5 panels): the pad rows are zeros at the input and are
        # carried through the row-wise kernels (LayerNorm, GEMMs) like any other row -- attention works per frame and never reads them -- so that no GEMM of
        # the trunk ends in a ragged tail (round 5: the 128-row tail launches behind the assembly GEMMs were 6.5 % of the ViT's kernel time)
        n_real = B * S
        n = (n_real + 255) // 256 * 256
        x = torch.empty(n, C, device=dev, dtype=BF16)
        if n > n_real:
            x[n_real:].zero_()
        ops.vit_tokens(pt, rt["cls"], rt["pos"], B, NP, C, x)
        ao = torch.empty(n, C, device=dev, dtype=BF16)      # attention output, reused by every block (attention writes the B * S real rows)
        if n > n_real:
            ao[n_real:].zero_()
        for b, w in zip(self.blocks, rt["blocks"]):
            h, _, _ = ops.norm_fwd(x, b.norm1.weight, b.norm1.bias, 1e-6, n, D=C, save_stats=False)
            qkv = ops.gemm_nt(h, w["qkv"], n, 3 * C, C, bias=w["qkv_b"])
            ops.attn_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], 3 * C, B, S, self.heads, 0.125, save_lse=False, out=ao)
            x = ops.gemm_nt(ao, w["proj"], n, C, C, bias=w["proj_b"], residual=x)
            h, _, _ = ops.norm_fwd(x, b.norm2.weight, b.norm2.bias, 1e-6, n, D=C, save_stats=False)
            f = ops.gemm_nt(h, w["fc1"], n, 4 * C, C, bias=w["fc1_b"], act=ops.ACT_GELU)
            x = ops.gemm_nt(f, w["fc2"], n, C, 4 * C, bias=w["fc2_b"], residual=x)
        out, _, _ = ops.norm_fwd(x, self.norm.weight, self.norm.bias, 1e-6, n_real, D=C, save_stats=False)
        return out[:n_real].view(B, S, C)


class _ViTPreprocessorBase:
    """Raw uint8 frames -> frozen ViT -> AdaptiveAvgPool2d((7, 12)).  ``process`` returns the reference's fp32 (B, C, 7, 12);
    ``process_tokens`` writes the storage-native bf16 tokens [B, ncam, 84, C]."""
    MEAN, STD, CROP_X, HW = DINO_RGB_MEANS, DINO_RGB_STDS, 3, (224, 384)

    def _setup(self, rgb_input_uuid, output_uuid, model_type, device, flatten):
        self.input_uuids, self.uuid, self.device = [rgb_input_uuid], output_uuid, torch.device(device)
        self.vit = DinoViT(self.device, **VIT_PRESETS[model_type])
        C = self.vit.dim
        self.observation_space = Box(-float("inf"), float("inf"), (7 * 12, C) if flatten else (7, 12, C))

    def to(self, device):
        return self

    def _tokens(self, fr):
        assert tuple(fr.shape[1:3]) == self.HW, f"Expected shape is {self.HW[0]}x{self.HW[1]}; got {tuple(fr.shape[1:3])}"
        return self.vit.patch_tokens(fr, self.MEAN, self.STD, crop_x=self.CROP_X)

    @torch.no_grad()
    def process(self, obs: Dict[str, torch.Tensor], *a, **k) -> torch.Tensor:
        fr = obs[self.input_uuids[0]].to(self.device)
        x = self._tokens(fr)
        B, v = fr.shape[0], self.vit
        gh, gw = self._rt_grid()
        out = torch.empty(B, v.dim, 7, 12, device=self.device, dtype=torch.float32)
        ops.adaptive_pool_tokens(x, B, 1 if v.has_cls else 0, gh, gw, v.dim, 7, 12, chw_out=out)
        return out

    def _rt_grid(self):
        return self.vit._rt["gh"], self.vit._rt["gw"]

    @torch.no_grad()
    def process_tokens(self, frames_u8: torch.Tensor, out_tokens: torch.Tensor, cam: int, ncam: int = 2):
        x = self._tokens(frames_u8.to(self.device))
        gh, gw = self._rt_grid()
        ops.adaptive_pool_tokens(x, frames_u8.shape[0], 1 if self.vit.has_cls else 0, gh, gw, self.vit.dim, 7, 12, cam=cam, ncam=ncam, tok_out=out_tokens)


    @torch.no_grad()
    def process_tokens_all_cameras(self, frames_u8: torch.Tensor, out_tokens: torch.Tensor):
        """frames_u8 [ncam * B, H, W, 3] (camera-major: all envs' frames of camera 0, then camera 1, ...) -> out_tokens [B, ncam, 84, C] in ONE pass of the trunk
        (the rollout's two cameras share the frozen encoder: one 2B-frame batch fills the GPU better than two B-frame batches)."""
        B, ncam = out_tokens.shape[0], out_tokens.shape[1]
        assert frames_u8.shape[0] == ncam * B
        x = self._tokens(frames_u8.to(self.device))
        gh, gw = self._rt_grid()
        for cam in range(ncam):
            ops.adaptive_pool_tokens(x[cam * B:(cam + 1) * B], B, 1 if self.vit.has_cls else 0, gh, gw, self.vit.dim, 7, 12, cam=cam, ncam=ncam, tok_out=out_tokens)


class DinoViTPreprocessor(_ViTPreprocessorBase):
    """dino_preprocessors.py:38-125: 224 x 384 frames, W crop [3:-3], DINOv2 ``x_norm_patchtokens`` -> (B, C, 16, 27) -> pool (7, 12)."""

    def __init__(self, rgb_input_uuid: str, output_uuid: str, dino_model_type: str = "dinov2_vits14", device="cuda", flatten: bool = True, **kw):
        if dino_model_type == "dinov2_vitg14":
            raise NotImplementedError("dinov2_vitg14 (SwiGLU-fused MLP, 40 blocks) is not built; ViT-S/B/L-14 are")
        assert dino_model_type in ("dinov2_vits14", "dinov2_vitb14", "dinov2_vitl14"), dino_model_type
        self._setup(rgb_input_uuid, output_uuid, dino_model_type, device, flatten)


class SigLIPPreprocessor(_ViTPreprocessorBase):
    """siglip_preprocessors.py:18-104: 256 x 256 frames, mean = std = 0.5, timm trunk ``forward_features`` (B, 256, 768) ->
    (B, 768, 16, 16) -> AdaptiveAvgPool2d((7, 12)).  Weights come from open_clip's hub download in the reference: random-init geometry
    here (parity pinned against the fp32 restatement of the published timm forward, oracle/ref_vit.py)."""
    SIGLIP_RGB_MEANS, SIGLIP_RGB_STDS = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)
    MEAN, STD, CROP_X, HW = SIGLIP_RGB_MEANS, SIGLIP_RGB_STDS, 0, (256, 256)

    def __init__(self, rgb_input_uuid: str, output_uuid: str, siglip_model_type: str = "ViT-B-16-SigLIP-256", device="cuda", flatten: bool = True, **kw):
        assert siglip_model_type in ("ViT-B-16-SigLIP-256", "ViT-L-16-SigLIP-256"), siglip_model_type
        self._setup(rgb_input_uuid, output_uuid, siglip_model_type, device, flatten)
