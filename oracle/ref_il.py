"""TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the imitation-learning model.

``EarlyFusionCnnTransformer`` with the llama decoder and 512-wide transformers
(/root/reference/architecture/models/transformer_models/early_fusion_tsfm_models.py:49-207,221-312) on pre-encoded image
features: the frozen image encoder is outside this restatement (its features are the input, as for the RL towers).  Same
state_dict names as the reference module minus ``visual_encoder.image_encoder.*``.  Pinned against the reference itself by
tests/golden/g8_il.npz (``small_3``) and tests/golden/g9_il_siglip.npz (``siglip_base_3``: the SigLIP text tower is third-party and enters
both sides as oracle.ref_siglip_text; the fixture pins the reference's code around it) -- tests/golden/make_golden_il.py.
"""
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ref_model import N_ACTIONS, RefGoalEncoder, RefLlamaDecoder, RefPositionalEncoder

NAV, MANIP = "raw_navigation_camera", "raw_manipulation_camera"


class RefEarlyFusion(nn.Module):
    def __init__(self, max_length=1000, max_batch=8, n_fusion_layers=3, n_decoder_layers=3, dino_dim=384, text_encoder="t5-small", d_model=512, n_heads=8, n_heads_decoder=None):
        """defaults = ``small_3``; (6, 6, 384) = ``small_6``; (3, 3, 768) = ``base_3``; (3, 3, 768, "SigLIPBase") = ``siglip_base_3``;
        (3, 3, 1024, "SigLIPLarge") = ``siglip_large_3``; (3, 3, 2048) = ``clip_resnet_50_3``; (6, 3, 768, "SigLIPBase", 768, 12) = ``siglip_base_6_3``
        (early_fusion_tsfm_models.py:221-312)."""
        super().__init__()
        d = d_model
        self.siglip = text_encoder != "t5-small"
        te, td = None, 512
        if self.siglip:
            from .ref_siglip_text import RefSigLIPText
            cfg = {"SigLIPBase": dict(width=768, heads=12, layers=12), "SigLIPLarge": dict(width=1024, heads=16, layers=24)}[text_encoder]
            te, td = RefSigLIPText(**cfg), cfg["width"]
            te.output_tokens = True                                                # text_cond_visual_encoder.py:39
        self.visual_encoder = RefGoalEncoder(tokenizer=None, d=d, n_heads=n_heads, n_layers=n_fusion_layers, dino_dim=dino_dim, text_encoder=te, text_dim=td)
        self.decoder = RefLlamaDecoder(d, n_decoder_layers, n_heads if n_heads_decoder is None else n_heads_decoder, 1e-5, max_batch, max_length)
        self.actor = nn.Linear(d, N_ACTIONS)
        self.time_encoder = RefPositionalEncoder(d)
        self.last_actions_embed = nn.Embedding(N_ACTIONS + 2, d, padding_idx=N_ACTIONS + 1)
        self.object_in_hand_embed = nn.Embedding(3, d)

    def encode(self, batch: Dict[str, torch.Tensor]):
        ve = self.visual_encoder
        nav = batch[NAV]
        B, T = nav.shape[:2]
        R = B * T
        with torch.no_grad():   # text_cond_visual_encoder.py:144-152
            if self.siglip:     # isinstance(..., TextTransformer): tokens then the pooled token (:146-148)
                cls_feats, text = ve.text_encoder(batch["goals"])
                text = torch.cat([text, cls_feats.unsqueeze(1)], dim=1)
            else:
                text = ve.text_encoder(batch["goals"]["input_ids"], batch["goals"]["attention_mask"])
        text = ve.text_adapter(text)                                             # [B, L, d]
        parts = [ve.fusion_token.view(1, 1, -1).expand(R, -1, -1)]
        for key, tok in sorted([(MANIP, ve.visual_sensor_token_raw_manipulation_camera), (NAV, ve.visual_sensor_token_raw_navigation_camera)]):
            parts.append(ve._camera(batch[key].reshape(R, *nav.shape[-3:]), tok))   # sensors in sorted order (:104)
        parts.append(text.unsqueeze(1).expand(B, T, -1, -1).reshape(R, text.shape[1], -1))
        fused = ve.fusion_xformer(torch.cat(parts, dim=1))[:, 0]
        return fused.view(B, T, -1)

    def forward(self, batch):
        x = self.encode(batch)
        x = x + self.last_actions_embed(batch["last_actions"])
        x = x + self.object_in_hand_embed(batch["an_object_is_in_hand"])
        x = x + self.time_encoder(batch["time_ids"])
        B, T = x.shape[:2]
        mask = torch.tril(torch.ones(T, T, dtype=torch.bool))[None, None].expand(B, 1, T, T)
        logits = self.actor(self.decoder(x, 0, mask))
        loss = F.cross_entropy(logits.reshape(-1, N_ACTIONS), batch["actions"].reshape(-1), ignore_index=-1)
        return dict(actions_logits=logits, actions_loss=loss, loss=loss)
