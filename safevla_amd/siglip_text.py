"""Frozen SigLIP text tower of the imitation-learning presets ``siglip_*``.

Reference: ``create_text_encoder`` (/root/reference/architecture/models/transformer_models/text_cond_visual_encoder.py:35-45) takes
``open_clip.create_model_from_pretrained("hf-hub:timm/ViT-B-16-SigLIP-256")[0].text`` with ``output_tokens = True``;
``encode_text`` (:143-151) calls it under ``no_grad`` on the tokenizer's ``[B, 64]`` ids (SigLipPreprocessor.process_goals,
preprocessors.py:334-343: no attention mask, the tokenizer pads to the context length) and concatenates ``[tokens, pooled]`` ->
``[B, 65, width]`` in front of the trainable ``text_adapter`` (``TEXT_ENCODER_DIMS``: 768 / 1024, :24-32).

The tower itself is third-party (open_clip ``TextTransformer``; not in the reference tree, weights are a hub download): its published
forward for the SigLIP configuration is restated -- token embedding + learned positions, pre-LN residual blocks
(``nn.MultiheadAttention`` with fused ``in_proj``, exact GELU MLP 4x), no causal and no padding mask (``no_causal_mask``), ``ln_final``
(eps 1e-6), ``pool_type = "last"`` and a biased linear ``text_projection`` of the pooled token -- with open_clip's ``state_dict`` names,
random-init geometry.  PARITY UNPINNED against open_clip proper; pinned against the fp32 restatement oracle/ref_siglip_text.py, and the
in-tree arithmetic around it (concatenation order, adapter widths) against the reference itself by tests/golden/g9_il_siglip.npz.

Runs on the frozen-ViT kernels (preproc.DinoViT): bf16 MFMA GEMMs with bias / GELU / residual epilogues, fused attention at S = 64,
LayerNorm rows of width 768 / 1024; once per unique goal.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .model import _NS

BF16, F32 = torch.bfloat16, torch.float32

# open_clip model configs ``ViT-B-16-SigLIP-256`` / ``ViT-L-16-SigLIP-256`` (text_cfg): context 64, vocabulary 32 000, LayerNorm eps 1e-6
SIGLIP_TEXT_PRESETS = {
    "SigLIPBase": dict(width=768, heads=12, layers=12),
    "SigLIPLarge": dict(width=1024, heads=16, layers=24),
}


class SigLIPTextFrozen(nn.Module):
    def __init__(self, device, width=768, heads=12, layers=12, vocab=32000, context=64, eps=1e-6, tokens_exclude_pooled=False):
        """``tokens_exclude_pooled``: open_clip 3.x (the reference pins open_clip_torch==3.2.0, requirements.txt:132) returns every ``ln_final`` row as
        ``tokens`` next to the pooled one (-> L + 1 text tokens); releases before the pooling refactor returned ``x[:, :-1]`` for ``pool_type = "last"``
        (-> L).  Default = the pinned release."""
        super().__init__()
        self.tokens_exclude_pooled = tokens_exclude_pooled
        assert width // heads == 64, "attention kernels: head_dim 64"
        self.device_, self.width, self.heads, self.context, self.eps = device, width, heads, context, eps
        self.output_tokens = True
        d = torch.device(device)
        P = lambda *s, sc: nn.Parameter((torch.randn(*s) * sc).to(d), requires_grad=False)
        ones = lambda n: nn.Parameter(torch.ones(n, device=d), requires_grad=False)
        W = width
        self.token_embedding = _NS(); self.token_embedding.weight = P(vocab, W, sc=0.02)
        self.positional_embedding = P(context, W, sc=0.01)
        self.transformer = _NS()
        self.transformer.resblocks = nn.ModuleList()
        for _ in range(layers):
            b = _NS()
            b.ln_1 = _NS(); b.ln_1.weight = ones(W); b.ln_1.bias = P(W, sc=0.02)
            b.attn = _NS()
            b.attn.in_proj_weight = P(3 * W, W, sc=1.0 / math.sqrt(W)); b.attn.in_proj_bias = P(3 * W, sc=0.02)
            b.attn.out_proj = _NS(); b.attn.out_proj.weight = P(W, W, sc=1.0 / math.sqrt(W)); b.attn.out_proj.bias = P(W, sc=0.02)
            b.ln_2 = _NS(); b.ln_2.weight = ones(W); b.ln_2.bias = P(W, sc=0.02)
            b.mlp = _NS(); b.mlp.c_fc = _NS(); b.mlp.c_proj = _NS()
            b.mlp.c_fc.weight = P(4 * W, W, sc=1.0 / math.sqrt(W)); b.mlp.c_fc.bias = P(4 * W, sc=0.02)
            b.mlp.c_proj.weight = P(W, 4 * W, sc=1.0 / math.sqrt(4 * W)); b.mlp.c_proj.bias = P(W, sc=0.02)
            self.transformer.resblocks.append(b)
        self.ln_final = _NS(); self.ln_final.weight = ones(W); self.ln_final.bias = P(W, sc=0.02)
        self.text_projection = _NS(); self.text_projection.weight = P(W, W, sc=1.0 / math.sqrt(W)); self.text_projection.bias = P(W, sc=0.02)
        self._rt = None

    def sync(self, dtype=BF16):
        """(re)build the bf16 runtime copies of the frozen weights (the tower runs in bf16 in both precisions of the policy: like the frozen ViT)."""
        f = lambda t: t.float().contiguous()
        h = lambda t: t.to(BF16).contiguous()
        rt = dict(pos=f(self.positional_embedding), proj=h(self.text_projection.weight), proj_b=f(self.text_projection.bias), blocks=[])
        for b in self.transformer.resblocks:
            rt["blocks"].append(dict(qkv=h(b.attn.in_proj_weight), qkv_b=f(b.attn.in_proj_bias), o=h(b.attn.out_proj.weight), o_b=f(b.attn.out_proj.bias),
                                     fc=h(b.mlp.c_fc.weight), fc_b=f(b.mlp.c_fc.bias), pr=h(b.mlp.c_proj.weight), pr_b=f(b.mlp.c_proj.bias)))
        self._rt = rt

    @torch.no_grad()
    def encode(self, ids: torch.Tensor, attn_mask=None, drop_seed=None, drop_p: float = 0.0, dtype=BF16, seed_dev=None, fused=None) -> torch.Tensor:
        """ids [U, L] int64 (device; L <= context, normally the tokenizer's 64) -> ``cat([tokens, pooled])`` as rows [U * out_tokens(L), width].
        ``attn_mask`` / dropout / ``fused`` arguments are those of ``T5Frozen.encode`` and are unused: the tower has neither padding, dropout nor RMSNorms."""
        if self._rt is None:
            self.sync()
        rt, W, H = self._rt, self.width, self.heads
        U, L = ids.shape
        assert L <= self.context, f"{L} goal tokens > context length {self.context}"
        n = U * L
        emb = ops.embed_gather(self.token_embedding.weight, ids.reshape(-1).contiguous(), dtype=BF16)
        x = torch.empty(n, W, device=ids.device, dtype=BF16)
        ops.vit_tokens(emb, None, rt["pos"], U, L, W, x)                      # + positional_embedding[:L]
        for b, w in zip(self.transformer.resblocks, rt["blocks"]):
            h, _, _ = ops.norm_fwd(x, b.ln_1.weight, b.ln_1.bias, self.eps, n, D=W, save_stats=False)
            qkv = ops.gemm_nt(h, w["qkv"], n, 3 * W, W, bias=w["qkv_b"])
            ao, _ = ops.attn_fwd(qkv, qkv[:, W:], qkv[:, 2 * W:], 3 * W, U, L, H, 0.125, save_lse=False)
            x = ops.gemm_nt(ao, w["o"], n, W, W, bias=w["o_b"], residual=x)
            h, _, _ = ops.norm_fwd(x, b.ln_2.weight, b.ln_2.bias, self.eps, n, D=W, save_stats=False)
            f = ops.gemm_nt(h, w["fc"], n, 4 * W, W, bias=w["fc_b"], act=ops.ACT_GELU)
            x = ops.gemm_nt(f, w["pr"], n, W, 4 * W, bias=w["pr_b"], residual=x)
        Lo = self.out_tokens(L)
        out = torch.empty(U, Lo, W, device=ids.device, dtype=BF16)
        if Lo == L + 1:
            # ln_final rows written straight into the [U, L + 1, W] layout (row map: groups of L rows, L + 1 apart); pooled = projection of the last one
            ops.norm_fwd(x, self.ln_final.weight, self.ln_final.bias, self.eps, n, D=W, save_stats=False, y=out, ymap=(L, L + 1, 0))
            ops.gemm_nt(out[:, L - 1], rt["proj"], U, W, W, bias=rt["proj_b"], out=out[:, L], lda=(L + 1) * W, ldc=(L + 1) * W)
        else:
            # tokens = rows 0 .. L - 2, pooled = projection of row L - 1 written in its place
            ops.norm_fwd(x, self.ln_final.weight, self.ln_final.bias, self.eps, n, D=W, save_stats=False, y=out)
            last = out[:, L - 1].contiguous()
            ops.gemm_nt(last, rt["proj"], U, W, W, bias=rt["proj_b"], out=out[:, L - 1], ldc=L * W)
        out = out.view(U * Lo, W)
        return out if dtype == BF16 else out.to(dtype)

    def out_tokens(self, L: int) -> int:
        """text tokens handed to the adapter for L input ids"""
        return L if self.tokens_exclude_pooled else L + 1
