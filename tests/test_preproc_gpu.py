"""Frozen-ViT preprocessor (rollout-time part of the policy forward): HIP path vs the fp32 oracle restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def pre():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from safevla_amd.preproc import DinoViTPreprocessor

    torch.manual_seed(0)
    p = DinoViTPreprocessor("rgb_raw", "rgb_dinov2", device=DEV)
    with torch.no_grad():   # non-trivial LayerScale / norm weights so that every parameter matters
        for b in p.vit.blocks:
            b.ls1.gamma.copy_(0.5 + 0.5 * torch.rand(384, device=DEV))
            b.ls2.gamma.copy_(0.5 + 0.5 * torch.rand(384, device=DEV))
            b.norm1.weight.copy_(1 + 0.1 * torch.randn(384, device=DEV))
        p.vit.pos_embed.mul_(10.0)
    p.vit.sync()
    return p


def test_normalize_matches_reference_formula(pre):
    from oracle import ref_vit
    from safevla_amd.preproc import DataAugmentationPreprocessor

    fr = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (3, 224, 384, 3), dtype=np.uint8))
    got = DataAugmentationPreprocessor("rgb_raw", "rgb", device=DEV).process({"rgb_raw": fr})
    assert torch.allclose(got.cpu(), ref_vit.normalize(fr), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,H,W,crop_x,P,gh,gw", [(3, 224, 384, 3, 14, 16, 27),      # DINOv2 ViT-S/14 on the 224 x 384 frames (crop 3:-3)
                                                  (2, 256, 256, 0, 16, 16, 16),      # SigLIP 256 x 256, 16 x 16 patches, no crop
                                                  (2, 30, 45, 3, 14, 2, 2),          # odd row pitch (135 B): every staged row has its own 0..3-byte lead
                                                  (1, 28, 29, 1, 14, 2, 2)])         # crop ends at the last byte of the buffer: byte-wise staging path
def test_patchify_u8_matches_numpy_im2col(B, H, W, crop_x, P, gh, gw):
    """svla_patchify_u8_bf16 (normalise + crop + im2col, dino_preprocessors.py:27-35,224-239) against a numpy restatement, incl. the geometries that
    exercise the aligned-dword row staging (round 5): k = c*P*P + ky*P + kx, zero padded to KP."""
    from safevla_amd import ops
    from safevla_amd.preproc import DINO_RGB_MEANS, DINO_RGB_STDS

    rs = np.random.RandomState(B * 1000 + W)
    fr = rs.randint(0, 256, (B, H, W, 3), dtype=np.uint8)
    K = 3 * P * P
    KP = (K + 31) // 32 * 32
    out = torch.full((B, gh * gw, KP), float("nan"), device=DEV, dtype=torch.bfloat16)
    ops.patchify_u8(torch.from_numpy(fr).to(DEV), DINO_RGB_MEANS, DINO_RGB_STDS, out, crop_x=crop_x, P=P, gh=gh, gw=gw)
    x = (fr.astype(np.float32) / 255.0 - np.array(DINO_RGB_MEANS, np.float32)) / np.array(DINO_RGB_STDS, np.float32)
    x = x[:, :gh * P, crop_x:crop_x + gw * P]                                       # [B, gh*P, gw*P, 3]
    x = x.reshape(B, gh, P, gw, P, 3).transpose(0, 1, 3, 5, 2, 4).reshape(B, gh * gw, K)      # (c, ky, kx)
    got = out.float().cpu().numpy()
    assert (got[:, :, K:] == 0).all()
    np.testing.assert_allclose(got[:, :, :K], x, rtol=2.0 ** -7, atol=1e-6)


def test_vit_features_vs_oracle(pre):
    from oracle import ref_vit

    fr = torch.from_numpy(np.random.RandomState(1).randint(0, 256, (2, 224, 384, 3), dtype=np.uint8))
    sd = {k: v.detach().float().cpu() for k, v in pre.vit.state_dict().items()}
    want_tok, want_pool = ref_vit.vit_features(sd, fr)
    got_tok = pre.vit.patch_tokens(fr.to(DEV)).float().cpu()
    e = (got_tok - want_tok).abs().max().item() / want_tok.abs().max().item()
    assert e < 4e-2, e
    got = pre.process({"rgb_raw": fr}).cpu()
    assert got.shape == (2, 384, 7, 12) and got.dtype == torch.float32
    e2 = (got - want_pool).abs().max().item() / want_pool.abs().max().item()
    assert e2 < 4e-2, e2
    tok = torch.zeros(2, 2, 84, 384, device=DEV, dtype=torch.bfloat16)
    pre.process_tokens(fr, tok, cam=1)
    assert (tok[:, 0] == 0).all()
    assert torch.allclose(tok[:, 1].float().cpu(), got.flatten(2).permute(0, 2, 1), atol=2e-2, rtol=2e-2)


def test_state_dict_has_dinov2_names(pre):
    names = set(pre.vit.state_dict())
    for k in ("cls_token", "pos_embed", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.11.ls2.gamma", "blocks.5.mlp.fc1.bias", "norm.weight"):
        assert k in names, k
    assert pre.vit.state_dict()["pos_embed"].shape == (1, 1370, 384)


# ---- geometry-generic ViT kernels: patch 14 / 16, width 384 / 768, 433 / 256 tokens ---------------------------------
def _perturb(vit):
    with torch.no_grad():
        for b in vit.blocks:
            if vit.has_ls:
                b.ls1.gamma.copy_(0.5 + 0.5 * torch.rand(vit.dim, device=DEV))
                b.ls2.gamma.copy_(0.5 + 0.5 * torch.rand(vit.dim, device=DEV))
            b.norm1.weight.copy_(1 + 0.1 * torch.randn(vit.dim, device=DEV))
        vit.pos_embed.mul_(10.0)
    vit._rt = None


def test_siglip_preprocessor_vs_oracle():
    """SigLIPPreprocessor (siglip_preprocessors.py:18-104): 256 x 256, mean = std = 0.5, ViT-B/16 trunk without class token,
    (B,256,768) -> (B,768,16,16) -> AdaptiveAvgPool2d((7,12))."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import ref_vit
    from safevla_amd.preproc import SigLIPPreprocessor

    torch.manual_seed(1)
    p = SigLIPPreprocessor("rgb_raw", "rgb_siglip", device=DEV)
    assert p.observation_space.shape == (84, 768) and "cls_token" not in p.vit.state_dict() and p.vit.state_dict()["pos_embed"].shape == (1, 256, 768)
    _perturb(p.vit)
    fr = torch.from_numpy(np.random.RandomState(2).randint(0, 256, (2, 256, 256, 3), dtype=np.uint8))
    sd = {k: v.detach().float().cpu() for k, v in p.vit.state_dict().items()}
    want_tok, want_pool = ref_vit.vit_features(sd, fr, heads=12, native_grid=16, patch=16, crop_x=0, mean=(0.5,) * 3, std=(0.5,) * 3)
    got = p.process({"rgb_raw": fr}).cpu()
    assert got.shape == (2, 768, 7, 12) and want_tok.shape == (2, 256, 768)
    e = (got - want_pool).abs().max().item() / want_pool.abs().max().item()
    assert e < 4e-2, e
    with pytest.raises(AssertionError):
        p.process({"rgb_raw": torch.zeros(1, 224, 384, 3, dtype=torch.uint8)})       # "Expected shape is 256x256"


def test_dinov2_base_geometry_vs_oracle():
    """dino_model_type='dinov2_vitb14' (dino_preprocessors.py:60-64): width 768, 12 heads, same 433-token grid."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import ref_vit
    from safevla_amd.preproc import DinoViTPreprocessor

    torch.manual_seed(2)
    p = DinoViTPreprocessor("rgb_raw", "rgb_dinov2", dino_model_type="dinov2_vitb14", device=DEV)
    assert p.observation_space.shape == (84, 768)
    _perturb(p.vit)
    fr = torch.from_numpy(np.random.RandomState(3).randint(0, 256, (1, 224, 384, 3), dtype=np.uint8))
    sd = {k: v.detach().float().cpu() for k, v in p.vit.state_dict().items()}
    _, want_pool = ref_vit.vit_features(sd, fr, heads=12)
    got = p.process({"rgb_raw": fr}).cpu()
    e = (got - want_pool).abs().max().item() / want_pool.abs().max().item()
    assert got.shape == (1, 768, 7, 12) and e < 4e-2, e
    with pytest.raises(NotImplementedError):
        DinoViTPreprocessor("rgb_raw", "x", dino_model_type="dinov2_vitg14", device=DEV)


def test_all_cameras_in_one_pass_equals_per_camera_passes(pre):
    """process_tokens_all_cameras (both cameras of all envs through the frozen trunk as one batch) fills the same storage-native token tensor as one
    process_tokens call per camera."""
    B = 3
    fr = torch.from_numpy(np.random.RandomState(5).randint(0, 256, (2 * B, 224, 384, 3), dtype=np.uint8)).to(DEV)
    a = torch.zeros(B, 2, 84, 384, device=DEV, dtype=torch.bfloat16)
    b = torch.zeros_like(a)
    pre.process_tokens(fr[:B], a, cam=0)
    pre.process_tokens(fr[B:], a, cam=1)
    pre.process_tokens_all_cameras(fr, b)
    assert torch.isfinite(b.float()).all() and b.float().abs().sum().item() > 0
    # the GEMM kernels chosen for 3 and 6 frames may differ (tile vs panel kernels): one bf16 rounding apart at most
    assert torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2)
