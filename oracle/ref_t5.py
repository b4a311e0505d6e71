"""TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the frozen T5-small *encoder*.

The reference calls HuggingFace ``T5EncoderModel.from_pretrained("t5-small")`` under ``no_grad``
(/root/reference/architecture/models/allenact_transformer_models/allenact_dino_transformer.py:506-508,
599-603).  ``transformers`` is a third-party dependency (requirements.txt), not in the reference tree;
this file restates the published T5 v1.0 encoder algorithm (Raffel et al. 2020; HF ``modeling_t5``):
pre-RMS-norm blocks, un-scaled dot-product attention with a learned bucketed relative-position bias
shared by all layers, ReLU feed-forward without biases, final RMS norm.  PINNED against
``transformers.T5EncoderModel`` (same state_dict names) by tests/test_oracle_t5.py.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def relative_position_bucket_bidirectional(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 128):
    """rel = key_pos - query_pos (int64) -> bucket id in [0, num_buckets)."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    a = rel.abs()
    max_exact = nb // 2
    small = a < max_exact
    big = max_exact + (
        torch.log(a.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)
    ).long()
    big = torch.minimum(big, torch.full_like(big, nb - 1))
    return out + torch.where(small, a, big)


class _T5Norm(nn.Module):
    def __init__(self, d, eps):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.eps = eps

    def forward(self, x):
        return self.weight * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps))


class _SelfAttention(nn.Module):
    def __init__(self, d, h, dk, has_bias_table, num_buckets):
        super().__init__()
        self.q = nn.Linear(d, h * dk, bias=False)
        self.k = nn.Linear(d, h * dk, bias=False)
        self.v = nn.Linear(d, h * dk, bias=False)
        self.o = nn.Linear(h * dk, d, bias=False)
        if has_bias_table:
            self.relative_attention_bias = nn.Embedding(num_buckets, h)
        self.h, self.dk = h, dk


class _Layer0(nn.Module):
    def __init__(self, d, h, dk, has_bias_table, num_buckets, eps):
        super().__init__()
        self.SelfAttention = _SelfAttention(d, h, dk, has_bias_table, num_buckets)
        self.layer_norm = _T5Norm(d, eps)


class _FF(nn.Module):
    def __init__(self, d, dff):
        super().__init__()
        self.wi = nn.Linear(d, dff, bias=False)
        self.wo = nn.Linear(dff, d, bias=False)


class _Layer1(nn.Module):
    def __init__(self, d, dff, eps):
        super().__init__()
        self.DenseReluDense = _FF(d, dff)
        self.layer_norm = _T5Norm(d, eps)


class _Block(nn.Module):
    def __init__(self, d, h, dk, dff, first, num_buckets, eps):
        super().__init__()
        self.layer = nn.ModuleList([_Layer0(d, h, dk, first, num_buckets, eps), _Layer1(d, dff, eps)])


class _Stack(nn.Module):
    def __init__(self, shared, d, h, dk, dff, n_layers, num_buckets, eps):
        super().__init__()
        self.embed_tokens = shared
        self.block = nn.ModuleList([_Block(d, h, dk, dff, i == 0, num_buckets, eps) for i in range(n_layers)])
        self.final_layer_norm = _T5Norm(d, eps)


class RefT5Encoder(nn.Module):
    """state_dict names equal HF ``T5EncoderModel`` (shared.weight, encoder.block.N.layer...)."""

    def __init__(self, vocab=32128, d=512, h=8, dk=64, dff=2048, n_layers=6, num_buckets=32, max_distance=128, eps=1e-6):
        super().__init__()
        self.shared = nn.Embedding(vocab, d)
        self.encoder = _Stack(self.shared, d, h, dk, dff, n_layers, num_buckets, eps)
        self.h, self.dk, self.num_buckets, self.max_distance = h, dk, num_buckets, max_distance

    def position_bias(self, L, device):
        pos = torch.arange(L, device=device)
        rel = pos[None, :] - pos[:, None]  # key - query
        bucket = relative_position_bucket_bidirectional(rel, self.num_buckets, self.max_distance)
        tab = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight  # (buckets, h)
        return tab[bucket].permute(2, 0, 1)  # (h, L, L)

    hash_seed = None    # tests: apply the counter-based dropout masks of include/svla.h (streams as in model.T5Frozen.encode)
    drop_p = 0.1

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        U, L = input_ids.shape
        if self.hash_seed is not None and self.training:
            from .ref_model import hash_dropout
            drop = lambda t, k, attn=0: hash_dropout(t, self.hash_seed, k, self.drop_p, attn)
        else:
            drop = lambda t, k, attn=0: t
        x = drop(self.shared(input_ids), 62)
        bias = self.position_bias(L, x.device)[None]  # (1,h,L,L)
        bias = bias + (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
        for bi, blk in enumerate(self.encoder.block):
            s0 = 64 + 4 * bi
            sa, ln0 = blk.layer[0].SelfAttention, blk.layer[0].layer_norm
            n = ln0(x)
            q = sa.q(n).view(U, L, self.h, self.dk).transpose(1, 2)
            k = sa.k(n).view(U, L, self.h, self.dk).transpose(1, 2)
            v = sa.v(n).view(U, L, self.h, self.dk).transpose(1, 2)
            p = drop(F.softmax(q @ k.transpose(-1, -2) + bias, dim=-1), s0, L)  # note: no 1/sqrt(dk) in T5
            x = x + drop(sa.o((p @ v).transpose(1, 2).reshape(U, L, self.h * self.dk)), s0 + 1)
            ff, ln1 = blk.layer[1].DenseReluDense, blk.layer[1].layer_norm
            x = x + drop(ff.wo(drop(F.relu(ff.wi(ln1(x))), s0 + 2)), s0 + 3)
        return drop(self.encoder.final_layer_norm(x), 63)
