"""TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the SigLIP text tower used by the IL model's ``siglip_*`` presets.

Reference call sites: ``create_text_encoder`` / ``encode_text`` (/root/reference/architecture/models/transformer_models/
text_cond_visual_encoder.py:35-45,143-151): ``open_clip.create_model_from_pretrained("hf-hub:timm/ViT-B-16-SigLIP-256")[0].text`` with
``output_tokens = True``, called on the tokenizer's id tensor, returning ``(pooled, tokens)``.  The tower is third-party (open_clip
``TextTransformer``, requirements pin open_clip_torch; not vendored, weights are a hub download): its published forward for the SigLIP
text configuration (context 64, vocabulary 32 000, width 768 / 12 heads / 12 layers -- 1024 / 16 / 24 for ViT-L --, ``no_causal_mask``,
``pool_type = "last"``, ``proj_bias``, LayerNorm eps 1e-6, exact GELU) is restated here with open_clip's module / state_dict names --
**parity unpinned** against open_clip proper.  tests/golden/make_golden_il.py installs this class as ``open_clip.transformer.TextTransformer``
when it imports the reference, so that the reference's own code around the tower (concatenation of tokens and pooled token, adapter widths,
the fusion transformer over 1 + 168 + 65 tokens) is what the G9 fixture pins.
"""
import torch
import torch.nn as nn


class _Mlp(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.c_fc = nn.Linear(w, 4 * w)
        self.gelu = nn.GELU()
        self.c_proj = nn.Linear(4 * w, w)

    def forward(self, x):
        return self.c_proj(self.gelu(self.c_fc(x)))


class _Block(nn.Module):
    def __init__(self, w, heads, eps):
        super().__init__()
        self.ln_1 = nn.LayerNorm(w, eps=eps)
        self.attn = nn.MultiheadAttention(w, heads, batch_first=True)
        self.ln_2 = nn.LayerNorm(w, eps=eps)
        self.mlp = _Mlp(w)

    def forward(self, x):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False)[0]
        return x + self.mlp(self.ln_2(x))


class _Stack(nn.Module):
    def __init__(self, w, heads, layers, eps):
        super().__init__()
        self.resblocks = nn.ModuleList([_Block(w, heads, eps) for _ in range(layers)])

    def forward(self, x):
        for b in self.resblocks:
            x = b(x)
        return x


class RefSigLIPText(nn.Module):
    def __init__(self, width=768, heads=12, layers=12, vocab=32000, context=64, eps=1e-6, tokens_exclude_pooled=False):
        """``tokens_exclude_pooled`` = False: open_clip 3.x (the pinned 3.2.0), ``tokens`` are all ``ln_final`` rows; True: the older
        ``text_global_pool`` that returned ``x[:, :-1]`` for ``pool_type = "last"``."""
        super().__init__()
        self.output_tokens = False
        self.tokens_exclude_pooled = tokens_exclude_pooled
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(context, width))
        self.transformer = _Stack(width, heads, layers, eps)
        self.ln_final = nn.LayerNorm(width, eps=eps)
        self.text_projection = nn.Linear(width, width, bias=True)

    def forward(self, text):
        L = text.shape[1]
        x = self.token_embedding(text) + self.positional_embedding[:L]
        x = self.ln_final(self.transformer(x))
        pooled, tokens = x[:, -1], (x[:, :-1] if self.tokens_exclude_pooled else x)            # pool_type "last"
        pooled = self.text_projection(pooled)
        return (pooled, tokens) if self.output_tokens else pooled
