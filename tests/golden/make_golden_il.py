#!/usr/bin/env python3
"""Golden vectors of the imitation-learning model, produced by the REFERENCE ITSELF (build container only).

Imports ``EarlyFusionCnnTransformer`` from /root/reference with ``sys.modules`` shims for what the image lacks (open_clip,
torchvision-heavy preprocessors, the hub-downloaded image encoder -> an identity stub: the inputs are pre-encoded DINOv2 features,
T5 from a config instead of the hub).  Weights are name-seeded (oracle.detfill) on both sides.  Emits tests/golden/g8_il.npz:
seeded batch, logits, loss and gradient checksums of every trained tensor (``small_3``), and tests/golden/g9_il_siglip.npz: the same for the
reference's ``siglip_base_3`` preset.  The SigLIP text tower is open_clip's (absent here, hub weights): the restatement
oracle/ref_siglip_text.py is installed as ``open_clip.transformer.TextTransformer`` / returned by ``create_model_from_pretrained``, so G9 pins
the reference's own code around the tower (``encode_text``'s [tokens, pooled] concatenation, the 768 -> 512 adapter, the fusion transformer over
1 + 168 + 65 tokens, the preset table), not the tower."""
import importlib
import os
import sys
import types
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.detfill import fill_state_dict, grad_probe  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path=None):
    m = _mod(name)
    m.__path__ = [path] if path else []
    return m


def install():
    oc = _pkg("open_clip")
    oc.create_model_from_pretrained = None
    _mod("open_clip.tokenizer", HFTokenizer=object)
    from oracle.ref_siglip_text import RefSigLIPText
    _mod("open_clip.transformer", TextTransformer=RefSigLIPText)
    for p in ("architecture", "architecture/models", "architecture/models/transformer_models", "training", "training/offline", "utils",
              "utils/constants"):
        _pkg(p.replace("/", "."), os.path.join(REF, p))

    @dataclass
    class _EncCfg:
        model: str = "stub"
        output_size: tuple = (384, 7, 12)

    class _IdentityEncoder(nn.Module):   # pre-encoded features in, features out
        def __init__(self, cfg):
            super().__init__()
            self.cfg = cfg

        def forward(self, x):
            return x

    import dataclasses
    import typing
    # early_fusion_tsfm_models.py relies on ``from ...image_encoders import *`` for torch / nn / dataclass
    _mod("architecture.models.transformer_models.image_encoders",
         IMAGE_ENCODERS={"Dinov2Small": (_IdentityEncoder, _EncCfg()),
                         "SigLIPBase": (_IdentityEncoder, _EncCfg(model="ViT-B-16-SigLIP-256", output_size=(768, 7, 12)))},
         torch=torch, nn=nn, dataclass=dataclasses.dataclass, SigLIP=type("SigLIP", (), {}), List=typing.List, np=np)
    _mod("architecture.models.transformer_models.preprocessors", Preprocessor=object, PreprocessorConfig=object, SigLipPreprocessor=object,
         SigLipPreprocessorConfig=object, tensor_image_preprocessor=None)
    _mod("training.offline.train_utils", load_pl_ckpt=None)
    _mod("utils.constants.stretch_initialization_utils", ALL_STRETCH_ACTIONS=[str(i) for i in range(20)])
    _mod("utils.nn_utils", create_causal_mask=lambda T, device: torch.triu(torch.full([T, T], float("-inf"), device=device), diagonal=1),
         sample_action_index_from_logits=None)
    sys.path.insert(0, REF)


def main():
    install()
    from transformers import T5Config, T5EncoderModel

    class _T5:
        @staticmethod
        def from_pretrained(name):
            return T5EncoderModel(T5Config(vocab_size=32128, d_model=512, d_kv=64, d_ff=2048, num_layers=6, num_heads=8, feed_forward_proj="relu"))

    tc = importlib.import_module("architecture.models.transformer_models.text_cond_visual_encoder")
    tc.T5EncoderModel = _T5
    ef = importlib.import_module("architecture.models.transformer_models.early_fusion_tsfm_models")
    cfg = ef.EarlyFusionCnnTransformerConfig()
    cfg.visual_encoder = tc.TextCondVisualEncoderConfig()
    cfg.visual_encoder.input_sensors = ["raw_navigation_camera", "raw_manipulation_camera", "last_actions", "an_object_is_in_hand"]
    cfg.decoder = tc.TransformerConfig(3, 512, 8)
    torch.manual_seed(0)
    model = ef.EarlyFusionCnnTransformer(cfg).eval()
    fill_state_dict(model, seed=7, share_t5=False)

    B, T, L = 2, 8, 9
    rs = np.random.RandomState(21)
    valid = np.array([8, 5])
    batch = {
        # fp16-representable values: stored as float16 in the fixture (half the bytes), used as float32 on both sides
        "raw_navigation_camera": rs.standard_normal((B, T, 384, 7, 12)).astype(np.float16).astype(np.float32),
        "raw_manipulation_camera": rs.standard_normal((B, T, 384, 7, 12)).astype(np.float16).astype(np.float32),
        "time_ids": np.tile(np.arange(T), (B, 1)).astype(np.int64),
        "an_object_is_in_hand": rs.randint(0, 3, size=(B, T)).astype(np.int64),
        "actions": rs.randint(0, 20, size=(B, T)).astype(np.int64),
    }
    ids = rs.randint(3, 32000, size=(B, L)).astype(np.int64)
    am = np.ones((B, L), np.int64)
    for b, n in enumerate([9, 5]):
        ids[b, n - 1] = 1; ids[b, n:] = 0; am[b, n:] = 0
    last = np.full((B, T), 21, np.int64)
    pad = np.zeros((B, T), bool)
    for b in range(B):
        last[b, 0] = 20
        last[b, 1:valid[b]] = batch["actions"][b, : valid[b] - 1]
        batch["actions"][b, valid[b]:] = -1
        pad[b, valid[b]:] = True
    batch["last_actions"], batch["padding_mask"] = last, pad
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    tb["goals"] = dict(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(am))
    for p in model.parameters():
        p.grad = None
    out = model(tb)
    out["loss"].backward()
    g = dict(batch)
    for k in ("raw_navigation_camera", "raw_manipulation_camera"):
        g[k] = g[k].astype(np.float16)
    g.update(goal_ids=ids, goal_mask=am, logits=out["actions_logits"].detach().numpy(), loss=np.float64(out["loss"].item()))
    names = []
    for n, p in model.named_parameters():
        if p.grad is not None and "text_encoder" not in n:
            g["gp:" + n] = np.array(grad_probe(n, p.grad), np.float64)
            names.append(n)
    np.savez_compressed(os.path.join(HERE, "g8_il.npz"), **g)
    with open(os.path.join(HERE, "state_dict_manifest_il.txt"), "w") as f:
        for k, v in model.state_dict().items():
            f.write(f"{k}\t{tuple(v.shape)}\n")
    print("wrote g8_il.npz: loss", out["loss"].item(), "trained tensors", len(names))
    main_siglip(tc, ef)


def main_siglip(tc, ef):
    """G9: the reference's ``siglip_base_3`` preset, configured as ``build_model`` configures it (early_fusion_tsfm_models.py:255-259; ``build_model``
    itself goes on to build the dataset preprocessor and to ``json.dump`` the dataclass configs into the working directory, :317-349, which is not
    part of the model)."""
    from oracle.ref_siglip_text import RefSigLIPText

    def fake_create(name):
        assert name == "hf-hub:timm/ViT-B-16-SigLIP-256", name
        return (types.SimpleNamespace(text=RefSigLIPText(width=768, heads=12, layers=12), context_length=64),)

    tc.create_model_from_pretrained = fake_create
    torch.manual_seed(0)
    cfg = ef.EarlyFusionCnnTransformerConfig()
    cfg.visual_encoder = tc.TextCondVisualEncoderConfig()
    cfg.visual_encoder.input_sensors = ["raw_navigation_camera", "raw_manipulation_camera", "last_actions", "an_object_is_in_hand"]
    cfg.visual_encoder.image_encoder = "SigLIPBase"
    cfg.visual_encoder.text_encoder = "SigLIPBase"
    cfg.visual_encoder.fusion_xformer = tc.TransformerConfig(3, 512, 8)
    cfg.decoder = tc.TransformerConfig(3, 512, 8)
    model = ef.EarlyFusionCnnTransformer(cfg).eval()
    assert isinstance(model.visual_encoder.text_encoder, tc.TextTransformer) and model.visual_encoder.text_encoder.output_tokens
    fill_state_dict(model, seed=11, share_t5=False)
    B, T, L = 2, 6, 64
    rs = np.random.RandomState(33)
    valid = np.array([6, 4])
    batch = {
        "raw_navigation_camera": rs.standard_normal((B, T, 768, 7, 12)).astype(np.float16).astype(np.float32),
        "raw_manipulation_camera": rs.standard_normal((B, T, 768, 7, 12)).astype(np.float16).astype(np.float32),
        "time_ids": np.tile(np.arange(T), (B, 1)).astype(np.int64),
        "an_object_is_in_hand": rs.randint(0, 3, size=(B, T)).astype(np.int64),
        "actions": rs.randint(0, 20, size=(B, T)).astype(np.int64),
    }
    ids = np.ones((B, L), np.int64)                               # the SigLIP tokenizer pads with 1 to the context length (preprocessors.py:339)
    for b, n in enumerate([11, 7]):
        ids[b, :n] = rs.randint(3, 32000, size=n)
    last = np.full((B, T), 21, np.int64)
    pad = np.zeros((B, T), bool)
    for b in range(B):
        last[b, 0] = 20
        last[b, 1:valid[b]] = batch["actions"][b, : valid[b] - 1]
        batch["actions"][b, valid[b]:] = -1
        pad[b, valid[b]:] = True
    batch["last_actions"], batch["padding_mask"] = last, pad
    tb = {k: torch.from_numpy(v) for k, v in batch.items()}
    tb["goals"] = torch.from_numpy(ids)
    for p in model.parameters():
        p.grad = None
    out = model(tb)
    out["loss"].backward()
    g = dict(batch)
    for k in ("raw_navigation_camera", "raw_manipulation_camera"):
        g[k] = g[k].astype(np.float16)
    g.update(goal_ids=ids, logits=out["actions_logits"].detach().numpy(), loss=np.float64(out["loss"].item()))
    names = []
    for n, p in model.named_parameters():
        if p.grad is not None and "text_encoder" not in n:
            g["gp:" + n] = np.array(grad_probe(n, p.grad), np.float64)
            names.append(n)
    np.savez_compressed(os.path.join(HERE, "g9_il_siglip.npz"), **g)
    with open(os.path.join(HERE, "state_dict_manifest_il_siglip.txt"), "w") as f:
        for k, v in model.state_dict().items():
            f.write(f"{k}\t{tuple(v.shape)}\n")
    print("wrote g9_il_siglip.npz: loss", out["loss"].item(), "trained tensors", len(names))


if __name__ == "__main__":
    main()
