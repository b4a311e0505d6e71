"""Host mirror of the reference's three-tower actor-critic, running on the gfx950 HIP kernels.

Drop-in surface (same names, argument meaning and state_dict keys as the reference):
  * ``SafeDinoLLAMATxNavActorCriticSeparate.forward(observations, memory, prev_actions, masks)``
        /root/reference/architecture/models/allenact_transformer_models/separate_actor_critic.py:22-37
  * one tower = ``DinoLLAMATxNavActorCritic``   .../allenact_dino_transformer.py:47-475
  * ``DinoTxGoalEncoder``                        .../allenact_dino_transformer.py:478-717
  * llama ``TransformerDecoder``                 /root/reference/training/online/third_party_models/llama/model.py:425-467
  * ``sampler_select``, ``recurrent_memory_specification`` (AllenAct ActorCriticModel API)

MI355X-first design (not a translation of the nn.Module graph):
  * all trainable parameters of the three towers live in ONE flat fp32 arena (plus flat grad / Adam state / bf16
    mirror): a single fused Adam+clip launch, a single RCCL all-reduce, weight-gradient GEMMs accumulate in place;
  * each tower's forward/backward is an explicit kernel schedule over bf16 activations (no autograd tracing inside);
    autograd sees one node per tower, so ``loss.backward()`` of the reference API still works;
  * the frozen T5 text encoder runs once per *unique goal string* (content-hashed on the GPU) instead of once per
    (step, env) row per tower; DINO features are re-laid out once per forward (fp32 channels-first -> bf16 tokens)
    and shared by the three towers.
No CPU / eager fallback: every op below is a C-ABI kernel (include/svla.h); missing library => import error.
"""
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

import torch
import torch.nn as nn

from . import ops
from .api import CategoricalDistr, SafeActorCriticOutput
from .losses import HLGaussLoss
from .text import GoalTokenizer, bytes_to_str

N_ACTIONS = 20
D = 512
DINO = 384
NPATCH = 84          # 7 x 12 grid per camera
TEXT_OFF = 1 + 2 * NPATCH
TEXT_ENCODER_DIMS = {"t5-small": 512, "SigLIPBase": 768, "SigLIPLarge": 1024}
BF16, F32 = torch.bfloat16, torch.float32
import os as _os0
_NO_COMPRESSOR_BITS = _os0.environ.get("SVLA_NO_COMPRESSOR_BITS", "0") == "1"     # A/B switch of the 1-bit compressor ReLU masks


class _NS(nn.Module):
    """bare namespace module (parameters are attached by the arena)"""


def _seq(n):
    return nn.Sequential(*[_NS() for _ in range(n)])


# ================================================================================================ parameter arena
class _Arena:
    """Flat fp32 parameter arena shared by the three towers."""

    def __init__(self):
        self.specs: List[Tuple[nn.Module, str, Tuple[int, ...], str]] = []
        self.offsets: Dict[int, Tuple[int, int]] = {}

    def declare(self, owner: nn.Module, name: str, shape, init: str):
        self.specs.append((owner, name, tuple(shape), init))

    def begin_tower(self, align: int = 256):
        """Start a new tower's contiguous range on an ``align``-element boundary (16-byte aligned bf16 weight views for the LDS-DMA
        loads of every tower -- one tower has an odd number of trainable elements -- and per-tower all-reduce / Adam ranges)."""
        n = sum(int(np.prod(s[2])) for s in self.specs)
        pad = (-n) % align
        if pad:
            self.specs.append((None, "_pad", (pad,), "zeros"))
        self._tower_starts = getattr(self, "_tower_starts", []) + [n + pad]

    def build(self, device):
        total = sum(int(np.prod(s[2])) for s in self.specs)
        total += (-total) % 256
        starts = getattr(self, "_tower_starts", [0])
        self.tower_ranges = [(a, b) for a, b in zip(starts, starts[1:] + [total])]
        self.total = total
        self.flat_p = torch.zeros(total, device=device, dtype=F32)
        self.flat_g = torch.zeros(total, device=device, dtype=F32)
        self.flat_m = torch.zeros(total, device=device, dtype=F32)
        self.flat_v = torch.zeros(total, device=device, dtype=F32)
        self.flat_bf16 = torch.zeros(total, device=device, dtype=BF16)
        off = 0
        g = torch.Generator().manual_seed(0)
        for owner, name, shape, init in self.specs:
            n = int(np.prod(shape))
            if owner is None:        # alignment padding: stays zero (zero gradient => Adam never moves it)
                off += n
                continue
            view = self.flat_p[off:off + n].view(shape)
            view.copy_(_init_tensor(shape, init, g).to(device))
            p = nn.Parameter(view, requires_grad=True)
            p.grad = self.flat_g[off:off + n].view(shape)
            setattr(owner, name, p)
            self.offsets[id(p)] = (off, n)
            off += n

    def slab(self, p: nn.Parameter, flat: torch.Tensor, shape=None):
        off, n = self.offsets[id(p)]
        return flat[off:off + n].view(shape if shape is not None else p.shape)


def _init_tensor(shape, init, g):
    if init == "ones":
        return torch.ones(shape)
    if init == "zeros":
        return torch.zeros(shape)
    if init == "tok":
        return 0.1 * torch.rand(shape, generator=g)
    if init == "emb":
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.01
    if init == "lin":
        fan_in = int(np.prod(shape[1:]))
        return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
    if init == "actor":
        w = torch.empty(shape)
        nn.init.orthogonal_(w, gain=0.01)
        return w
    if init == "critic":
        w = torch.empty(shape)
        nn.init.orthogonal_(w)
        return w
    raise KeyError(init)


# ================================================================================================ one tower
class Tower(nn.Module):
    """``DinoLLAMATxNavActorCritic`` (full-sensor configuration of dinov2_vits_tsfm_base.py:234-270)."""

    def __init__(self, arena: _Arena, device, n_fusion_layers=3, n_decoder_layers=3, max_steps=500, critic_type="linear",
                 precision="bf16", dino_dim=DINO, text_encoder="t5-small", d_model=512, n_heads=8, n_heads_decoder=None):
        super().__init__()
        # transformer width of the fusion encoder AND the decoder (512 x 8 heads everywhere in the RL towers, allenact_dino_transformer.py:101-117; the imitation-
        # learning presets also use 768 x 12, early_fusion_tsfm_models.py:275-294); llama's SwiGLU hidden size follows from it (llama_model.py:330-334)
        n_heads_decoder = n_heads if n_heads_decoder is None else n_heads_decoder
        if d_model % 64 or d_model > 1024 or d_model % n_heads or d_model % n_heads_decoder or max(d_model // n_heads, d_model // n_heads_decoder) > 128:
            raise NotImplementedError(f"transformer width {d_model} with {n_heads} / {n_heads_decoder} heads")
        D = self.D = d_model
        H = self.H = n_heads                      # fusion transformer (nn.TransformerEncoderLayer nhead)
        self.Hdec = n_heads_decoder               # llama decoder (ModelArgs.n_heads); TransformerConfig(3, 768, 8) + TransformerConfig(6, 768, 12) = siglip_base_3_6
        # head widths: 64 everywhere on the MFMA attention kernels; any other width (96 = 768 / 8: base_6, siglip_base_3_6) takes the fp32 attention kernels
        # through fp32 copies of the operands (ops.attn_fwd: a slow path)
        self.hdim, self.hdim_dec = d_model // n_heads, d_model // n_heads_decoder
        HD = self.dec_hidden = 256 * ((int(2 * 4 * d_model / 3) + 255) // 256)
        self.dino_dim = dino_dim          # channel width of the frozen image features: 384 (ViT-S/14), 768 (ViT-B/14, SigLIP-B), 1024 (ViT-L), 2048 (CLIP RN50)
        # frozen text encoder and the width of its features (text_cond_visual_encoder.py:24-45 ``TEXT_ENCODER_DIMS`` / ``create_text_encoder``): the RL towers
        # and the t5 presets of the IL model use t5-small; the IL model's ``siglip_*`` presets the SigLIP text tower (tokens + pooled token, siglip_text.py)
        if text_encoder not in TEXT_ENCODER_DIMS:
            raise NotImplementedError("Only SigLIP and T5 text encoders are supported.")
        self.text_encoder_name, self.text_dim = text_encoder, TEXT_ENCODER_DIMS[text_encoder]
        if precision not in ("bf16", "fp32"):
            raise ValueError(f"precision must be 'bf16' (MFMA product path) or 'fp32' (verification mode), got {precision!r}")
        # activation / GEMM-operand dtype.  "fp32" = the verification mode: the same schedule on the fp32 twins of every kernel
        # (include/svla.h, last section), for comparing with the reference's fp32 arithmetic at fp32 tolerance; not a fast path.
        self.precision, self.adt = precision, (F32 if precision == "fp32" else BF16)
        if critic_type not in ("linear", "mlp", "discrete"):
            print(f"Unknown critic type: {critic_type}")
            raise NotImplementedError
        self.critic_type = critic_type
        self.arena = arena
        self.device_ = device
        self.max_steps = max_steps
        self.time_step_counter = 0
        self.prune_last = True      # dead-output elimination in the last fusion layer (exact; see run_forward)
        # BASELINE config 5 ("fp8 MFMA attention"): the full-sequence fusion-encoder attention layers run on the e4m3 / e5m2 kernels
        # (svla_attn_fp8_*); the pruned last layer (one query per row) and the decoder keep the bf16 kernels.  bf16 precision only.
        self.fp8_attention = False
        # The reference leaves the policy in train() mode (allenact_dino_transformer.py:193): nn.TransformerEncoderLayer's
        # dropout 0.1 is active in rollouts and updates.  Same here: ``.eval()`` turns it off (parity fixtures are eval-mode).
        self.dropout_p = 0.1
        self.t5_dropout = True      # the frozen text encoder's own dropout (also active in the reference's train mode)
        self.drop_seed_base = 0x5AFE + 977 * len(arena.specs)   # distinct per tower; settable for reproducible tests
        self._fwd_count = 0
        dec = arena.declare
        ve = self.visual_encoder = _NS()
        dec(ve, "fusion_token", (D,), "tok")
        dec(ve, "visual_sensor_token_raw_navigation_camera", (D,), "tok")   # adjacent: [2, 512] camera-token table
        dec(ve, "visual_sensor_token_raw_manipulation_camera", (D,), "tok")
        ve.text_adapter = _seq(3)
        dec(ve.text_adapter[0], "weight", (D, self.text_dim), "lin"); dec(ve.text_adapter[0], "bias", (D,), "zeros")
        dec(ve.text_adapter[1], "weight", (D,), "ones"); dec(ve.text_adapter[1], "bias", (D,), "zeros")
        ve.visual_compressor = _seq(4)
        dec(ve.visual_compressor[0], "weight", (D, dino_dim, 1, 1), "lin"); dec(ve.visual_compressor[0], "bias", (D,), "zeros")
        dec(ve.visual_compressor[2], "weight", (D, D, 1, 1), "lin"); dec(ve.visual_compressor[2], "bias", (D,), "zeros")
        ve.visual_adapter = _seq(3)
        dec(ve.visual_adapter[0], "weight", (D, D), "lin"); dec(ve.visual_adapter[0], "bias", (D,), "zeros")
        dec(ve.visual_adapter[1], "weight", (D,), "ones"); dec(ve.visual_adapter[1], "bias", (D,), "zeros")
        ve.fusion_xformer = _NS()
        ve.fusion_xformer.layers = nn.ModuleList()
        for _ in range(n_fusion_layers):
            l = _NS()
            l.self_attn = _NS()
            dec(l.self_attn, "in_proj_weight", (3 * D, D), "lin"); dec(l.self_attn, "in_proj_bias", (3 * D,), "zeros")
            l.self_attn.out_proj = _NS()
            dec(l.self_attn.out_proj, "weight", (D, D), "lin"); dec(l.self_attn.out_proj, "bias", (D,), "zeros")
            l.linear1 = _NS(); dec(l.linear1, "weight", (2048, D), "lin"); dec(l.linear1, "bias", (2048,), "zeros")
            l.linear2 = _NS(); dec(l.linear2, "weight", (D, 2048), "lin"); dec(l.linear2, "bias", (D,), "zeros")
            l.norm1 = _NS(); dec(l.norm1, "weight", (D,), "ones"); dec(l.norm1, "bias", (D,), "zeros")
            l.norm2 = _NS(); dec(l.norm2, "weight", (D,), "ones"); dec(l.norm2, "bias", (D,), "zeros")
            ve.fusion_xformer.layers.append(l)
        if text_encoder == "t5-small":
            ve.text_encoder = T5Frozen(device)
        else:
            from .siglip_text import SIGLIP_TEXT_PRESETS, SigLIPTextFrozen
            ve.text_encoder = SigLIPTextFrozen(device, **SIGLIP_TEXT_PRESETS[text_encoder])
        self.object_in_hand_embed = _NS(); dec(self.object_in_hand_embed, "weight", (3, D), "emb")
        self.last_actions_embed = _NS(); dec(self.last_actions_embed, "weight", (N_ACTIONS + 2, D), "emb")
        self.time_encoder = _NS()
        self.time_encoder.register_buffer("div_term", torch.exp(torch.arange(0, D, 2) * (-math.log(10000.0) / D)).to(device))
        self.decoder = _NS()
        self.decoder.layers = nn.ModuleList()
        for _ in range(n_decoder_layers):
            l = _NS()
            l.attention = _NS()
            for n in ("wq", "wk", "wv"):                                   # adjacent: fused [3 D, D] QKV weight
                setattr(l.attention, n, _NS()); dec(getattr(l.attention, n), "weight", (D, D), "lin")
            l.attention.wo = _NS(); dec(l.attention.wo, "weight", (D, D), "lin")
            l.feed_forward = _NS()
            l.feed_forward.w1 = _NS(); l.feed_forward.w2 = _NS(); l.feed_forward.w3 = _NS()
            dec(l.feed_forward.w1, "weight", (HD, D), "lin")             # w1, w3 adjacent: fused [2 HD, D]
            dec(l.feed_forward.w3, "weight", (HD, D), "lin")
            dec(l.feed_forward.w2, "weight", (D, HD), "lin")
            l.attention_norm = _NS(); dec(l.attention_norm, "weight", (D,), "ones")
            l.ffn_norm = _NS(); dec(l.ffn_norm, "weight", (D,), "ones")
            self.decoder.layers.append(l)
        self.decoder.norm = _NS(); dec(self.decoder.norm, "weight", (D,), "ones")
        self.decoder.output = _NS(); dec(self.decoder.output, "weight", (D, D), "lin")
        self.actor = _NS(); self.actor.linear = _NS()
        dec(self.actor.linear, "weight", (N_ACTIONS, D), "actor"); dec(self.actor.linear, "bias", (N_ACTIONS,), "zeros")
        # critic heads of allenact_dino_transformer.py:147-162 (LinearCriticHead [3P] / MLPCriticHead :720-740 / DiscreteCriticHead :743-766)
        self.critic = _NS()
        if critic_type == "linear":
            self.critic.fc = _NS()
            dec(self.critic.fc, "weight", (1, D), "critic"); dec(self.critic.fc, "bias", (1,), "zeros")
            self._head_dims = []
        else:
            dims = [D, 256, 101] if critic_type == "discrete" else [D, 256, 256, 1]
            self.critic.fc = _seq(2 * (len(dims) - 1) - 1)              # Linear, ReLU, Linear[, ReLU, Linear]: state_dict keys fc.0 / fc.2 [/ fc.4]
            for j in range(len(dims) - 1):
                dec(self.critic.fc[2 * j], "weight", (dims[j + 1], dims[j]), "critic"); dec(self.critic.fc[2 * j], "bias", (dims[j + 1],), "zeros")
            self._head_dims = dims
            if critic_type == "discrete":   # "bins = 101 ... -5 to 15 ... sigma=0.15" (:152-156)
                self.critic.loss_fn = HLGaussLoss(min_value=-5.0, max_value=15.0, num_bins=101, sigma=0.15)
        self._wt: Dict[str, torch.Tensor] = {}

    # ---- weight views -------------------------------------------------------------------------------------
    def _gemm_weights(self):
        """(key, parameter(s), [N, K]) for every MFMA GEMM weight of the tower."""
        D, H, HD = self.D, self.H, self.dec_hidden
        ve = self.visual_encoder
        out = [("c1", [ve.visual_compressor[0].weight], (D, self.dino_dim)), ("c2", [ve.visual_compressor[2].weight], (D, D)),
               ("va", [ve.visual_adapter[0].weight], (D, D)), ("ta", [ve.text_adapter[0].weight], (D, self.text_dim))]
        for i, l in enumerate(ve.fusion_xformer.layers):
            out += [(f"f{i}.in", [l.self_attn.in_proj_weight], (3 * D, D)), (f"f{i}.out", [l.self_attn.out_proj.weight], (D, D)),
                    (f"f{i}.l1", [l.linear1.weight], (2048, D)), (f"f{i}.l2", [l.linear2.weight], (D, 2048))]
        for i, l in enumerate(self.decoder.layers):
            a, f = l.attention, l.feed_forward
            out += [(f"d{i}.qkv", [a.wq.weight, a.wk.weight, a.wv.weight], (3 * D, D)), (f"d{i}.wo", [a.wo.weight], (D, D)),
                    (f"d{i}.w13", [f.w1.weight, f.w3.weight], (2 * HD, D)), (f"d{i}.w2", [f.w2.weight], (D, HD))]
        out.append(("dout", [self.decoder.output.weight], (D, D)))
        return out

    def bind(self):
        """Create bf16 / fp32-grad views into the arena (call once after arena.build)."""
        D, H, HD = self.D, self.H, self.dec_hidden
        ar = self.arena
        self._w, self._dw = {}, {}
        for key, ps, (n, k) in self._gemm_weights():
            off, _ = ar.offsets[id(ps[0])]
            self._w[key] = (ar.flat_p if self.adt == F32 else ar.flat_bf16)[off:off + n * k].view(n, k)
            self._dw[key] = ar.flat_g[off:off + n * k].view(n, k)
            self._wt[key] = torch.empty(k, n, device=self.device_, dtype=self.adt)
        ve = self.visual_encoder
        off, _ = ar.offsets[id(ve.visual_sensor_token_raw_navigation_camera)]
        self._camtok = ar.flat_p[off:off + 2 * D].view(2, D)
        self._dcamtok = ar.flat_g[off:off + 2 * D].view(2, D)

    def refresh_transposes(self):
        ar = self.arena
        for key, ps, (n, k) in self._gemm_weights():
            off, _ = ar.offsets[id(ps[0])]
            ops.transpose_cast_bf16(ar.flat_p[off:off + n * k].view(n, k), self._wt[key])
        self._wg_dirty = True

    def refresh_folded(self):
        """W * gamma[None, :] (bf16) of the llama decoder's pre-norm linears -- RMSNorm(x) @ W^T = rstd(x) * (x @ (W gamma)^T) -- for the acting step's norm-fused
        GEMMs (ops.gemm_nt_rmsa).  Persistent buffers refreshed IN PLACE (recorded acting steps hold their addresses), lazily: only an acting step after an
        optimiser step / load_state_dict pays for it."""
        D, H, HD = self.D, self.H, self.dec_hidden
        if self.adt != BF16 or not getattr(self, "_wg_dirty", True):
            return
        ar = self.arena
        if not hasattr(self, "_wg"):
            self._wg = {}
        for i, l in enumerate(self.decoder.layers):
            a, f = l.attention, l.feed_forward
            for key, first, n, gamma in ((f"d{i}.qkv", a.wq.weight, 3 * D, l.attention_norm.weight), (f"d{i}.w13", f.w1.weight, 2 * HD, l.ffn_norm.weight)):
                off, _ = ar.offsets[id(first)]
                w32 = ar.flat_p[off:off + n * D].view(n, D)
                if key not in self._wg:
                    self._wg[key] = torch.empty(n, D, device=self.device_, dtype=BF16)
                self._wg[key].copy_(w32 * gamma.detach().float()[None, :])
        self._wg_dirty = False

    def _drop_sites(self):
        """Dropout descriptors of one forward pass: site(layer, k), k = 0 attention probabilities, 1 attention sub-layer output,
        2 feed-forward activation, 3 feed-forward sub-layer output; None in eval mode.  The pass seed is saved with the
        activations so that the backward regenerates the same masks."""
        if not self.training or self.dropout_p <= 0:
            return None, (lambda i, k, rm=1: None)
        self._fwd_count += 1
        seed = (self.drop_seed_base * 0x9E3779B1 + self._fwd_count * 0x85EBCA77) & 0xFFFFFFFF
        return seed, self._site_fn(seed)

    def _site_fn(self, seed):
        p = self.dropout_p
        sd = getattr(self, "_seed_dev", None)      # captured acting graph: the pass seed lives in device memory (fresh noise per replay)
        return (lambda i, k, rm=1: ops.Dropout(seed, 4 * i + k, p, rm, seed_dev=sd)) if seed is not None else (lambda i, k, rm=1: None)

    def g(self, p):  # fp32 grad view of a parameter
        return self.arena.slab(p, self.arena.flat_g)

    # ---- forward ----------------------------------------------------------------------------------------------
    # ---- acting path state (llama KV caches, llama/model.py:224-247; counter semantics allenact_dino_transformer.py:376-406)
    def _ensure_caches(self, B: int):
        D, H, HD = self.D, self.H, self.dec_hidden
        if getattr(self, "_kv", None) is None or self._kv[0].shape[0] < B:
            self._kv_version = getattr(self, "_kv_version", 0) + 1        # recorded steps hold pointers into the caches: new caches, new plans
            self._kv = [torch.zeros(B, self.max_steps, 2 * D, device=self.device_, dtype=self.adt) for _ in self.decoder.layers]

    def cache_select(self, keep: list):
        if getattr(self, "_kv", None) is not None:
            idx = torch.as_tensor(keep, device=self.device_, dtype=torch.long)
            self._kv = [c[idx].contiguous() for c in self._kv]
            self._kv_version = getattr(self, "_kv_version", 0) + 1

    def run_forward(self, prep: "Prep", need_grad: bool):
        D, H, HD = self.D, self.H, self.dec_hidden
        SCF, SCD = self.hdim ** -0.5, self.hdim_dec ** -0.5          # 0.125 for 64-wide heads
        T, B, R, S, L, U = prep.T, prep.B, prep.R, prep.S, prep.L, prep.U
        if T > 1 or need_grad or not getattr(prep, "acting", True) or self.time_step_counter >= self.max_steps:
            self.time_step_counter = 0
        ve, w = self.visual_encoder, self._w
        M2, M = R * 2 * NPATCH, R * S
        c = {}  # saved activations
        c["drop_seed"], site = self._drop_sites()
        tok = prep.tokens.view(M2, self.dino_dim)
        # ReLU derivatives of the two compressor convs as 1 bit per element (like the feed-forward's): the input-gradient GEMMs then read
        # M2 x 64 mask bytes instead of re-reading a whole M2 x 512 bf16 activation
        bits = need_grad and self.adt == BF16 and not _NO_COMPRESSOR_BITS
        c1b = torch.empty(ops.relu_bits_bytes(M2, D), device=self.device_, dtype=torch.uint8) if bits else None
        c2b = torch.empty(ops.relu_bits_bytes(M2, D), device=self.device_, dtype=torch.uint8) if bits else None
        c1 = ops.gemm_nt(tok, w["c1"], M2, D, self.dino_dim, bias=ve.visual_compressor[0].bias, act=ops.ACT_RELU, relu_bits_out=c1b)
        c2 = ops.gemm_nt(c1, w["c2"], M2, D, D, bias=ve.visual_compressor[2].bias, act=ops.ACT_RELU, relu_bits_out=c2b)
        a1 = ops.gemm_nt(c2, w["va"], M2, D, D, bias=ve.visual_adapter[0].bias)
        x = torch.empty(R, S, D, device=self.device_, dtype=self.adt)
        _, va_mean, va_rstd = ops.norm_fwd(a1, ve.visual_adapter[1].weight, ve.visual_adapter[1].bias, 1e-5, M2, relu=True,
                                           tok=self._camtok, tok_group=NPATCH, y=x, ymap=(2 * NPATCH, S, 1), D=D)
        t5_seed = c["drop_seed"] if self.t5_dropout else None
        key = getattr(prep, "ids_key", None)
        if t5_seed is None and key is not None and getattr(self, "_t5_cache", (None, None))[0] == key:
            t5 = self._t5_cache[1]     # eval mode: the frozen encoder is a pure function of the goal tokens (one episode = one goal)
        else:
            t5 = ve.text_encoder.encode(prep.ids, getattr(prep, "attn_mask_u8", None) if getattr(prep, "attn_mask_u8", None) is not None else prep.attn_mask,
                                        drop_seed=t5_seed, drop_p=self.dropout_p, dtype=self.adt,
                                        seed_dev=getattr(self, "_seed_dev", None), fused=getattr(self, "t5_fused", None))   # [U*L, text_dim], frozen (SigLIP: ids are [U, L - 1], the pooled token is row L - 1)
            self._t5_cache = (key, t5) if (t5_seed is None and key is not None) else (None, None)
        ta = ops.gemm_nt(t5, w["ta"], U * L, D, self.text_dim, bias=ve.text_adapter[0].bias)
        tf, ta_mean, ta_rstd = ops.norm_fwd(ta, ve.text_adapter[1].weight, ve.text_adapter[1].bias, 1e-5, U * L, relu=True, D=D)
        ops.fusion_fill(ve.fusion_token, tf, prep.gid, x, R, S, L, TEXT_OFF)
        c.update(c1=c1, c2=c2, c1b=c1b, c2b=c2b, a1=a1, va=(va_mean, va_rstd), t5=t5, ta=ta, ta_stats=(ta_mean, ta_rstd))
        xf = x.view(M, D)
        fl = []
        nfl = len(ve.fusion_xformer.layers)
        xf_stride = S * D
        for i, l in enumerate(ve.fusion_xformer.layers):
            if i == nfl - 1 and self.prune_last:
                # Only sequence position 0 of the last fusion layer is consumed (allenact_dino_transformer.py:708 x[:, 0]):
                # K/V are still projected for all tokens, but Q / attention / out_proj / norm1 / FFN / norm2 run on the R
                # position-0 rows only.  Identical outputs and gradients; ~84 % less work in this layer.
                b_in = l.self_attn.in_proj_bias
                kv = ops.gemm_nt(xf, w[f"f{i}.in"][D:], M, 2 * D, D, bias=b_in[D:])
                q0 = ops.gemm_nt(xf, w[f"f{i}.in"][:D], R, D, D, bias=b_in[:D], lda=S * D)
                ao, lse = ops.attn_fwd(q0, kv, kv[:, D:], 2 * D, R, S, H, SCF, save_lse=need_grad, Sq=1, ldq=D, drop=site(i, 0), head_dim=self.hdim)
                h1 = ops.gemm_nt(ao, w[f"f{i}.out"], R, D, D, bias=l.self_attn.out_proj.bias, residual=xf, ldr=S * D, drop=site(i, 1, S))
                x1, m1, r1 = ops.norm_fwd(h1, l.norm1.weight, l.norm1.bias, 1e-5, R, save_stats=need_grad, D=D)
                f1 = ops.gemm_nt(x1, w[f"f{i}.l1"], R, 2048, D, bias=l.linear1.bias, act=ops.ACT_RELU, drop=site(i, 2, S))
                h2 = ops.gemm_nt(f1, w[f"f{i}.l2"], R, D, 2048, bias=l.linear2.bias, residual=x1, drop=site(i, 3, S))
                xo, m2, r2 = ops.norm_fwd(h2, l.norm2.weight, l.norm2.bias, 1e-5, R, save_stats=need_grad, D=D)
                if need_grad:
                    fl.append(dict(pruned=True, x=xf, kv=kv, q0=q0, ao=ao, lse=lse, h1=h1, x1=x1, n1=(m1, r1), f1=f1, h2=h2, n2=(m2, r2)))
                xf, xf_stride = xo, D
                continue
            qkv = ops.gemm_nt(xf, w[f"f{i}.in"], M, 3 * D, D, bias=l.self_attn.in_proj_bias)
            f8 = None
            if self.fp8_attention and self.adt == BF16 and S <= 256 and self.hdim == 64:
                f8 = ops.attn_fp8_quant(qkv, 3 * D, R, S, H)
                ao, lse = ops.attn_fp8_fwd(f8, SCF, save_lse=need_grad, drop=site(i, 0))
                qkv = None                       # the backward reads the e4m3 copies
            else:
                ao, lse = ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, R, S, H, SCF, save_lse=need_grad, drop=site(i, 0), head_dim=self.hdim)
            h1 = ops.gemm_nt(ao, w[f"f{i}.out"], M, D, D, bias=l.self_attn.out_proj.bias, residual=xf, drop=site(i, 1))
            x1, m1, r1 = ops.norm_fwd(h1, l.norm1.weight, l.norm1.bias, 1e-5, M, save_stats=need_grad, D=D)
            # the ReLU derivative is kept as 1 bit per element (M x 256 bytes): the input-gradient GEMM then reads 16x fewer mask bytes
            f1b = torch.empty(ops.relu_bits_bytes(M, 2048), device=x1.device, dtype=torch.uint8) if (need_grad and self.adt == BF16) else None
            f1 = ops.gemm_nt(x1, w[f"f{i}.l1"], M, 2048, D, bias=l.linear1.bias, act=ops.ACT_RELU, relu_bits_out=f1b, drop=site(i, 2))
            h2 = ops.gemm_nt(f1, w[f"f{i}.l2"], M, D, 2048, bias=l.linear2.bias, residual=x1, drop=site(i, 3))
            xo, m2, r2 = ops.norm_fwd(h2, l.norm2.weight, l.norm2.bias, 1e-5, M, save_stats=need_grad, D=D)
            if need_grad:
                fl.append(dict(pruned=False, x=xf, qkv=qkv, f8=f8, ao=ao, lse=lse, h1=h1, x1=x1, n1=(m1, r1), f1=f1, f1b=f1b, h2=h2, n2=(m2, r2)))
            xf = xo
        c["fusion"] = fl
        # decoder over the rollout time axis, rows (b*T + t)
        j = torch.empty(R, D, device=self.device_, dtype=self.adt)
        ops.decoder_embed_fwd(xf, xf_stride, self.last_actions_embed.weight, self.object_in_hand_embed.weight,
                              self.time_encoder.div_term, prep.prev_actions, prep.masks, prep.hand, prep.time_step, T, B, j)
        xd = j
        dl = []
        if T == 1 and not need_grad and getattr(prep, "acting", True):
            # acting: one new token per env against the KV cache; env b attends to cache slots >= max(counter - time_step_b, 0)
            # (its current episode), allenact_dino_transformer.py:388-397.  A one-step UPDATE batch (need_grad: engine / fused-loss path on a T = 1 rollout)
            # takes the sequence branch below instead -- a sequence of length one has no history to attend to, and the backward needs the saved activations --
            # and so does a batch whose Prep says ``acting = False`` (the imitation-learning model's ``forward(batch)`` on one-step windows: independent
            # sequences, never the cache of an earlier call; its online agent sets ``acting = True``)
            if self.hdim_dec != 64:
                raise NotImplementedError("KV-cached single steps need 64-wide decoder heads (MFMA / decode attention kernels)")
            t = self.time_step_counter
            self._ensure_caches(B)
            t_dev = getattr(self, "_t_dev", None)
            if t_dev is None:
                start = torch.clamp(t - prep.time_step, min=0)
                kvalid = (torch.arange(t + 1, device=self.device_)[None, :] >= start[:, None]).to(torch.uint8).contiguous()
                S_att = t + 1
            else:
                # captured / recorded form: the step counter lives in device memory, so every kernel argument is step-independent --
                # attention runs over the whole cache window and the mask hides the slots beyond the counter
                kvalid = getattr(prep, "kvalid_static", None)        # shared by the three towers (a function of the step and time_step only)
                if kvalid is None:
                    ar = self._ar_steps
                    kvalid = ((ar[None, :] <= t_dev) & (ar[None, :] >= torch.clamp(t_dev - prep.time_step, min=0)[:, None])).to(torch.uint8).contiguous()
                S_att = self.max_steps
            # bf16 product path: RMSNorm folded into the following GEMM (one launch instead of two).  The update's sequence branch runs norm -> bf16 -> GEMM (its backward
            # needs the normed activation), so a rollout's old log-probs and the update's recomputed ones differ by that one rounding of the decoder's normed rows --
            # bounded by tests/test_model_gpu.py::test_fused_rmsnorm_step_close_to_unfused; ``rms_fused = False`` runs the acting step on the two-launch form
            fused = self.adt == BF16 and getattr(self, "rms_fused", True)
            if fused:
                self.refresh_folded()
            for i, l in enumerate(self.decoder.layers):
                if fused:
                    qkv = ops.gemm_nt_rmsa(xd, self._wg[f"d{i}.qkv"], B, 3 * D, D, 1e-5)
                else:
                    n1, _, _ = ops.norm_fwd(xd, l.attention_norm.weight, None, 1e-5, B, rms=True, save_stats=False, D=D)
                    qkv = ops.gemm_nt(n1, w[f"d{i}.qkv"], B, 3 * D, D)
                cache = self._kv[i]
                if t_dev is None:
                    cache[:B, t].copy_(qkv[:, D:])
                elif self.adt == BF16:
                    ops.kv_append(qkv[:, D:], 3 * D, cache, t_dev, B, 2 * D)
                else:
                    cache[:B].index_copy_(1, t_dev.view(1), qkv[:, D:].unsqueeze(1))
                cv = cache.view(-1, 2 * D)
                ao, _ = ops.attn_fwd(qkv, cv, cv[:, D:], 2 * D, B, S_att, self.Hdec, SCD, kvalid=kvalid, save_lse=False, Sq=1, ldq=3 * D,
                                     kv_rows=self.max_steps)
                h = ops.gemm_nt(ao, w[f"d{i}.wo"], B, D, D, residual=xd)
                if fused:
                    ab = ops.gemm_nt_rmsa(h, self._wg[f"d{i}.w13"], B, 2 * HD, D, 1e-5)
                else:
                    n2, _, _ = ops.norm_fwd(h, l.ffn_norm.weight, None, 1e-5, B, rms=True, save_stats=False, D=D)
                    ab = ops.gemm_nt(n2, w[f"d{i}.w13"], B, 2 * HD, D)
                gg = ops.swiglu_fwd(ab, B, HD)
                xd = ops.gemm_nt(gg, w[f"d{i}.w2"], B, D, HD, residual=h)
            self.time_step_counter += 1
        else:
          for i, l in enumerate(self.decoder.layers):
            n1, _, r1 = ops.norm_fwd(xd, l.attention_norm.weight, None, 1e-5, R, rms=True, save_stats=need_grad, D=D)
            qkv = ops.gemm_nt(n1, w[f"d{i}.qkv"], R, 3 * D, D)
            ao, lse = ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B, T, self.Hdec, SCD, head_dim=self.hdim_dec, mask_mode=ops.MASK_BLOCK_CAUSAL,
                                   traj=prep.traj_bt, save_lse=need_grad)
            h = ops.gemm_nt(ao, w[f"d{i}.wo"], R, D, D, residual=xd)
            n2, _, r2 = ops.norm_fwd(h, l.ffn_norm.weight, None, 1e-5, R, rms=True, save_stats=need_grad, D=D)
            ab = ops.gemm_nt(n2, w[f"d{i}.w13"], R, 2 * HD, D)
            gg = ops.swiglu_fwd(ab, R, HD)
            xo = ops.gemm_nt(gg, w[f"d{i}.w2"], R, D, HD, residual=h)
            if need_grad:
                dl.append(dict(x=xd, n1=n1, r1=r1, qkv=qkv, ao=ao, lse=lse, h=h, n2=n2, r2=r2, ab=ab, g=gg))
            xd = xo
        nf, _, rf = ops.norm_fwd(xd, self.decoder.norm.weight, None, 1e-5, R, rms=True, save_stats=need_grad, D=D)
        beliefs = ops.gemm_nt(nf, w["dout"], R, D, D, out_f32=True)             # fp32, rows (b*T + t)
        logits = ops.small_linear_fwd(beliefs, self.actor.linear.weight, self.actor.linear.bias, T, B)   # rows (t*B + b)
        full_logits = None
        if self.critic_type == "linear":
            values = ops.small_linear_fwd(beliefs, self.critic.fc.weight, self.critic.fc.bias, T, B)
        else:
            values, full_logits, c["head"] = self._critic_head_fwd(beliefs, T, B)
        c.update(dec=dl, xd_last=xd, nf=nf, rf=rf, beliefs=beliefs, xf_last=xf)
        self._last_full_logits = full_logits        # [T, B, 101] fp32 (critic_type == "discrete"), read by the 3-tower wrapper / engine
        return logits.view(T, B, N_ACTIONS), values.view(T, B, 1), (c if need_grad else None)

    # ---- MLP / discrete critic heads (fp32, strided-GEMM kernel; rows come in decoder order b*T + t, leave in (t*B + b)) --------
    def _critic_head_fwd(self, beliefs, T, B):
        R, dims, fc = T * B, self._head_dims, self.critic.fc
        hs = [beliefs]
        for j in range(len(dims) - 1):
            last = j == len(dims) - 2
            hs.append(ops.gemm_f32(hs[-1], fc[2 * j].weight, R, dims[j + 1], dims[j], bias=fc[2 * j].bias, act=ops.ACT_NONE if last else ops.ACT_RELU))
        out_tb = hs[-1].view(B, T, dims[-1]).transpose(0, 1).contiguous()           # -> rows (t*B + b)
        if self.critic_type == "discrete":
            values, _, _ = ops.hlgauss_fwd_bwd(out_tb.view(R, dims[-1]), None, None, self.critic.loss_fn.min_value, self.critic.loss_fn.max_value,
                                               self.critic.loss_fn.sigma, want_grad=False)
            return values, out_tb, hs
        return out_tb.reshape(R), None, hs

    def _critic_head_bwd(self, hs, T, B, dvalues, dfull, dbel, accumulate_dx):
        """dvalues [T,B,1] and/or dfull [T,B,101] -> parameter grads (arena) and dbel [R,512] (rows b*T + t)."""
        R, dims, fc, g = T * B, self._head_dims, self.critic.fc, self.g
        if self.critic_type == "discrete":
            lf = self.critic.loss_fn
            flog_tb = hs[-1].view(B, T, dims[-1]).transpose(0, 1).contiguous().view(R, dims[-1])
            d_tb = None
            if dvalues is not None:      # gradient through the read-out value = sum softmax * centres
                _, d_tb, _ = ops.hlgauss_fwd_bwd(flog_tb, None, dvalues.reshape(R).contiguous(), lf.min_value, lf.max_value, lf.sigma, want_values=False)
            if dfull is not None:
                d_tb = dfull.reshape(R, dims[-1]) if d_tb is None else d_tb + dfull.reshape(R, dims[-1])
        else:
            d_tb = dvalues.reshape(R, 1)
        dy = d_tb.view(T, B, dims[-1]).transpose(0, 1).contiguous().view(R, dims[-1]).float()     # back to decoder order
        for j in reversed(range(len(dims) - 1)):
            n_out, n_in, x = dims[j + 1], dims[j], hs[j]
            w = fc[2 * j].weight
            ops.gemm_f32(dy, x, n_out, n_in, R, sa=(1, n_out), sb=(1, n_in), out=g(w), accumulate=True)      # dW += dY^T X
            ops.colsum_f32(dy, g(fc[2 * j].bias), R, n_out)
            if j == 0:
                ops.gemm_f32(dy, w, R, n_in, n_out, sb=(1, n_in), out=dbel, accumulate=accumulate_dx)       # dX = dY W
            else:
                dy = ops.gemm_f32(dy, w, R, n_in, n_out, sb=(1, n_in), mask=x)                              # ... through the ReLU

    # ---- backward -----------------------------------------------------------------------------------------------
    def run_backward(self, prep: "Prep", c, dlogits: Optional[torch.Tensor], dvalues: Optional[torch.Tensor],
                     dfull_logits: Optional[torch.Tensor] = None):
        """Accumulates parameter gradients into the arena's flat grad buffer.  ``dfull_logits``: gradient of the discrete critic's
        bin logits (HL-Gauss loss), critic_type == "discrete" only."""
        D, H, HD = self.D, self.H, self.dec_hidden
        SCF, SCD = self.hdim ** -0.5, self.hdim_dec ** -0.5          # 0.125 for 64-wide heads
        T, B, R, S, L, U = prep.T, prep.B, prep.R, prep.S, prep.L, prep.U
        ve, w, wt, dw, g = self.visual_encoder, self._w, self._wt, self._dw, self.g
        M2, M = R * 2 * NPATCH, R * S
        dev = self.device_
        dbel = torch.empty(R, D, device=dev, dtype=F32)
        first = True
        if dlogits is not None:
            ops.small_linear_bwd(c["beliefs"], self.actor.linear.weight, dlogits.reshape(R, N_ACTIONS).contiguous(), dbel,
                                 g(self.actor.linear.weight), g(self.actor.linear.bias), T, B, accumulate_dx=False)
            first = False
        if self.critic_type != "linear":
            if dvalues is not None or dfull_logits is not None:
                self._critic_head_bwd(c["head"], T, B, dvalues, dfull_logits, dbel, accumulate_dx=not first)
                first = False
        elif dvalues is not None:
            ops.small_linear_bwd(c["beliefs"], self.critic.fc.weight, dvalues.reshape(R, 1).contiguous(), dbel,
                                 g(self.critic.fc.weight), g(self.critic.fc.bias), T, B, accumulate_dx=not first)
            first = False
        if first:
            return
        dy = torch.empty(R, D, device=dev, dtype=self.adt)
        ops.cast_bf16(dbel, dy)
        ops.gemm_tn_acc(dy, c["nf"], dw["dout"], R, D, D)
        dnf = ops.gemm_nt(dy, wt["dout"], R, D, D)
        dx = ops.norm_bwd(dnf, c["xd_last"], self.decoder.norm.weight, None, None, c["rf"], R, g(self.decoder.norm.weight), None, rms=True, D=D)
        for i in reversed(range(len(self.decoder.layers))):
            l, a = self.decoder.layers[i], c["dec"][i]
            ops.gemm_tn_acc(dx, a["g"], dw[f"d{i}.w2"], R, D, HD)
            dg = ops.gemm_nt(dx, wt[f"d{i}.w2"], R, HD, D)
            dab = ops.swiglu_bwd(a["ab"], dg, R, HD)
            ops.gemm_tn_acc(dab, a["n2"], dw[f"d{i}.w13"], R, 2 * HD, D)
            dn2 = ops.gemm_nt(dab, wt[f"d{i}.w13"], R, D, 2 * HD)
            dh = ops.norm_bwd(dn2, a["h"], l.ffn_norm.weight, None, None, a["r2"], R, g(l.ffn_norm.weight), None, rms=True, dres=dx, D=D)
            ops.gemm_tn_acc(dh, a["ao"], dw[f"d{i}.wo"], R, D, D)
            dao = ops.gemm_nt(dh, wt[f"d{i}.wo"], R, D, D)
            dqkv = torch.empty(R, 3 * D, device=dev, dtype=self.adt)
            q = a["qkv"]
            ops.attn_bwd(q, q[:, D:], q[:, 2 * D:], 3 * D, a["ao"], D, a["lse"], dao, D, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D,
                         B, T, self.Hdec, SCD, mask_mode=ops.MASK_BLOCK_CAUSAL, traj=prep.traj_bt, head_dim=self.hdim_dec)
            ops.gemm_tn_acc(dqkv, a["n1"], dw[f"d{i}.qkv"], R, 3 * D, D)
            dn1 = ops.gemm_nt(dqkv, wt[f"d{i}.qkv"], R, D, 3 * D)
            dx = ops.norm_bwd(dn1, a["x"], l.attention_norm.weight, None, None, a["r1"], R, g(l.attention_norm.weight), None, rms=True, dres=dh, D=D)
        pruned = bool(c["fusion"]) and c["fusion"][-1]["pruned"]
        if pruned:
            dxf = torch.empty(R, D, device=dev, dtype=self.adt)        # gradient of the position-0 outputs only
            ops.decoder_embed_bwd(dx, prep.prev_actions, prep.masks, prep.hand, T, B, dxf, D, g(self.last_actions_embed.weight),
                                  g(self.object_in_hand_embed.weight))
            dyf = dxf
        else:
            dxf = ops.zeros(R, S, D, device=dev, dtype=self.adt)
            ops.decoder_embed_bwd(dx, prep.prev_actions, prep.masks, prep.hand, T, B, dxf, S * D, g(self.last_actions_embed.weight),
                                  g(self.object_in_hand_embed.weight))
            dyf = dxf.view(M, D)
        site = self._site_fn(c.get("drop_seed"))
        drop_scale = 1.0 / (1.0 - self.dropout_p) if c.get("drop_seed") is not None else 1.0
        def masked_like(t, on):   # second norm_bwd output: the gradient of the dropped-out sub-layer output
            return torch.empty_like(t) if on else None
        for i in reversed(range(len(ve.fusion_xformer.layers))):
            l, a = ve.fusion_xformer.layers[i], c["fusion"][i]
            if a["pruned"]:
                d3, d1 = site(i, 3, S), site(i, 1, S)
                df = masked_like(a["h2"], d3 is not None)
                dh2 = ops.norm_bwd(dyf, a["h2"], l.norm2.weight, l.norm2.bias, a["n2"][0], a["n2"][1], R, g(l.norm2.weight), g(l.norm2.bias),
                                   dx_drop=df, drop=d3, D=D)
                df = dh2 if df is None else df           # grad of linear2's output (through dropout2); dh2 = residual-path grad
                ops.gemm_tn_acc(df, a["f1"], dw[f"f{i}.l2"], R, D, 2048, db=g(l.linear2.bias))
                # f1 is stored after ReLU and dropout: f1 > 0 <=> (pre-activation > 0 and kept); alpha = the dropout scale
                df1 = ops.gemm_nt(df, wt[f"f{i}.l2"], R, 2048, D, relu_mask=a["f1"], alpha=drop_scale)
                ops.gemm_tn_acc(df1, a["x1"], dw[f"f{i}.l1"], R, 2048, D, db=g(l.linear1.bias))
                dx1 = ops.gemm_nt(df1, wt[f"f{i}.l1"], R, D, 2048, residual=dh2)
                da = masked_like(a["h1"], d1 is not None)
                dh1 = ops.norm_bwd(dx1, a["h1"], l.norm1.weight, l.norm1.bias, a["n1"][0], a["n1"][1], R, g(l.norm1.weight), g(l.norm1.bias),
                                   dx_drop=da, drop=d1, D=D)
                da = dh1 if da is None else da
                ops.gemm_tn_acc(da, a["ao"], dw[f"f{i}.out"], R, D, D, db=g(l.self_attn.out_proj.bias))
                dao = ops.gemm_nt(da, wt[f"f{i}.out"], R, D, D)
                dq0 = torch.empty(R, D, device=dev, dtype=self.adt)
                dkv = torch.empty(M, 2 * D, device=dev, dtype=self.adt)
                kv = a["kv"]
                ops.attn_bwd(a["q0"], kv, kv[:, D:], 2 * D, a["ao"], D, a["lse"], dao, D, dq0, dkv, dkv[:, D:], 2 * D, R, S, H, SCF, head_dim=self.hdim,
                             Sq=1, ldq=D, lddq=D, drop=site(i, 0))
                gb = g(l.self_attn.in_proj_bias)
                ops.gemm_tn_acc(dkv, a["x"], dw[f"f{i}.in"][D:], M, 2 * D, D, db=gb[D:])
                ops.gemm_tn_acc(dq0, a["x"], dw[f"f{i}.in"][:D], R, D, D, ldx=S * D, db=gb[:D])
                dyf = ops.gemm_nt(dkv, wt[f"f{i}.in"][:, D:], M, D, 2 * D)                       # dX through K and V, all tokens
                t0 = ops.gemm_nt(dq0, wt[f"f{i}.in"][:, :D], R, D, D, residual=dh1)             # position 0: Q path + residual path
                ops.rows_add(dyf, S * D, t0, D, R, D)
                c["fusion"][i] = None
                continue
            d3, d1 = site(i, 3), site(i, 1)
            df = masked_like(a["h2"], d3 is not None)
            dh2 = ops.norm_bwd(dyf, a["h2"], l.norm2.weight, l.norm2.bias, a["n2"][0], a["n2"][1], M, g(l.norm2.weight), g(l.norm2.bias),
                               dx_drop=df, drop=d3, D=D)
            df = dh2 if df is None else df               # grad of linear2's output (through dropout2); dh2 = residual-path grad
            ops.gemm_tn_acc(df, a["f1"], dw[f"f{i}.l2"], M, D, 2048, db=g(l.linear2.bias))
            # the sign bits were taken after ReLU and dropout: bit <=> (pre-activation > 0 and kept); alpha = the dropout scale
            if a["f1b"] is not None:
                df1 = ops.gemm_nt(df, wt[f"f{i}.l2"], M, 2048, D, relu_bits=a["f1b"], alpha=drop_scale)
            else:       # fp32 verification mode: the stored activation itself is the mask
                df1 = ops.gemm_nt(df, wt[f"f{i}.l2"], M, 2048, D, relu_mask=a["f1"], alpha=drop_scale)
            ops.gemm_tn_acc(df1, a["x1"], dw[f"f{i}.l1"], M, 2048, D, db=g(l.linear1.bias))
            dx1 = ops.gemm_nt(df1, wt[f"f{i}.l1"], M, D, 2048, residual=dh2)
            del df1, df
            da = masked_like(a["h1"], d1 is not None)
            dh1 = ops.norm_bwd(dx1, a["h1"], l.norm1.weight, l.norm1.bias, a["n1"][0], a["n1"][1], M, g(l.norm1.weight), g(l.norm1.bias),
                               dx_drop=da, drop=d1, D=D)
            da = dh1 if da is None else da
            ops.gemm_tn_acc(da, a["ao"], dw[f"f{i}.out"], M, D, D, db=g(l.self_attn.out_proj.bias))
            dao = ops.gemm_nt(da, wt[f"f{i}.out"], M, D, D)
            dqkv = torch.empty(M, 3 * D, device=dev, dtype=self.adt)
            q = a["qkv"]
            if a.get("f8") is not None:
                ops.attn_fp8_bwd(a["f8"], a["ao"], a["lse"], dao, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, SCF, drop=site(i, 0))
            else:
                ops.attn_bwd(q, q[:, D:], q[:, 2 * D:], 3 * D, a["ao"], D, a["lse"], dao, D, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D,
                             R, S, H, SCF, drop=site(i, 0), head_dim=self.hdim)
            ops.gemm_tn_acc(dqkv, a["x"], dw[f"f{i}.in"], M, 3 * D, D, db=g(l.self_attn.in_proj_bias))
            dyf = ops.gemm_nt(dqkv, wt[f"f{i}.in"], M, D, 3 * D, residual=dh1)
            c["fusion"][i] = None
        dx0 = dyf
        ops.colsum_acc(dx0, g(ve.fusion_token), R, D, row_stride=S)
        # text adapter (trainable) -- the T5 encoder is frozen (no_grad in the reference)
        dtf = ops.zeros(U * L, D, device=dev, dtype=F32)
        if ops.det_active():       # deterministic mode: this accumulated intermediate gets its own fixed-point shadow (slot 1)
            dtf_sh = torch.zeros(U * L * D, device=dev, dtype=torch.int64)
            ops.det_config(1, dtf, dtf_sh)
            try:
                ops.fusion_text_bwd(dx0, prep.gid, T, B, S, L, TEXT_OFF, dtf)
                ops.det_finalize(dtf, dtf_sh)
            finally:
                ops.det_config(1, None, None)
        else:
            ops.fusion_text_bwd(dx0, prep.gid, T, B, S, L, TEXT_OFF, dtf)
        dtf_b = torch.empty(U * L, D, device=dev, dtype=self.adt)
        ops.cast_bf16(dtf, dtf_b)
        dta = ops.norm_bwd(dtf_b, c["ta"], ve.text_adapter[1].weight, ve.text_adapter[1].bias, c["ta_stats"][0], c["ta_stats"][1], U * L,
                           g(ve.text_adapter[1].weight), g(ve.text_adapter[1].bias), relu=True, D=D)
        ops.gemm_tn_acc(dta, c["t5"], dw["ta"], U * L, D, self.text_dim, db=g(ve.text_adapter[0].bias))
        # visual adapter + compressor (both cameras in one batch)
        da1 = ops.norm_bwd(dx0, c["a1"], ve.visual_adapter[1].weight, ve.visual_adapter[1].bias, c["va"][0], c["va"][1], M2,
                           g(ve.visual_adapter[1].weight), g(ve.visual_adapter[1].bias), relu=True, dtok=self._dcamtok,
                           tok_group=NPATCH, dymap=(2 * NPATCH, S, 1), D=D)
        ops.gemm_tn_acc(da1, c["c2"], dw["va"], M2, D, D, db=g(ve.visual_adapter[0].bias))
        dc2 = ops.gemm_nt(da1, wt["va"], M2, D, D, relu_bits=c["c2b"]) if c.get("c2b") is not None else ops.gemm_nt(da1, wt["va"], M2, D, D, relu_mask=c["c2"])
        ops.gemm_tn_acc(dc2, c["c1"], dw["c2"], M2, D, D, db=g(ve.visual_compressor[2].bias))
        dc1 = ops.gemm_nt(dc2, wt["c2"], M2, D, D, relu_bits=c["c1b"]) if c.get("c1b") is not None else ops.gemm_nt(dc2, wt["c2"], M2, D, D, relu_mask=c["c1"])
        ops.gemm_tn_acc(dc1, prep.tokens.view(M2, self.dino_dim), dw["c1"], M2, D, self.dino_dim, db=g(ve.visual_compressor[0].bias))


# ================================================================================================ frozen T5
class T5Frozen(nn.Module):
    """Frozen HF ``T5EncoderModel('t5-small')`` geometry (allenact_dino_transformer.py:506-508,599-603); state_dict
    names equal HF's.  Runs on the same bf16 GEMM / attention / RMS-norm kernels, once per unique goal."""

    def __init__(self, device, vocab=32128, d=512, h=8, dff=2048, n_layers=6, buckets=32, max_distance=128):
        super().__init__()
        self.device_ = device
        self.h, self.buckets, self.max_distance = h, buckets, max_distance

        def P(*shape, scale=None):
            t = torch.randn(*shape) * (scale if scale is not None else 1.0 / math.sqrt(shape[-1]))
            return nn.Parameter(t.to(device), requires_grad=False)

        self.shared = _NS(); self.shared.weight = P(vocab, d, scale=1.0)
        self.encoder = _NS()
        self.encoder.embed_tokens = _NS(); self.encoder.embed_tokens.weight = self.shared.weight   # tied, as in HF
        self.encoder.block = nn.ModuleList()
        for i in range(n_layers):
            b = _NS(); b.layer = nn.ModuleList([_NS(), _NS()])
            sa = b.layer[0].SelfAttention = _NS()
            for n in ("q", "k", "v", "o"):
                setattr(sa, n, _NS()); getattr(sa, n).weight = P(d, d)
            if i == 0:
                sa.relative_attention_bias = _NS(); sa.relative_attention_bias.weight = P(buckets, h, scale=0.5)
            b.layer[0].layer_norm = _NS(); b.layer[0].layer_norm.weight = nn.Parameter(torch.ones(d, device=device), requires_grad=False)
            ff = b.layer[1].DenseReluDense = _NS()
            ff.wi = _NS(); ff.wi.weight = P(dff, d)
            ff.wo = _NS(); ff.wo.weight = P(d, dff)
            b.layer[1].layer_norm = _NS(); b.layer[1].layer_norm.weight = nn.Parameter(torch.ones(d, device=device), requires_grad=False)
            self.encoder.block.append(b)
        self.encoder.final_layer_norm = _NS()
        self.encoder.final_layer_norm.weight = nn.Parameter(torch.ones(d, device=device), requires_grad=False)
        self._rt = None
        self._bias_cache: Dict[int, torch.Tensor] = {}

    def sync(self, dtype=BF16):
        """(re)build the runtime copies (bf16, or fp32 in the verification mode) of the frozen weights."""
        rt = []
        for b in self.encoder.block:
            sa, ff = b.layer[0].SelfAttention, b.layer[1].DenseReluDense
            qkv32 = torch.cat([sa.q.weight, sa.k.weight, sa.v.weight], 0).float()
            g0, g1 = b.layer[0].layer_norm.weight.float(), b.layer[1].layer_norm.weight.float()
            rt.append(dict(qkv=qkv32.to(dtype).contiguous(),
                           o=sa.o.weight.to(dtype).contiguous(), wi=ff.wi.weight.to(dtype).contiguous(),
                           wo=ff.wo.weight.to(dtype).contiguous(),
                           # T5LayerNorm folded into the following linear for the norm-fused small-M GEMMs (ops.gemm_nt_rmsa): W * gamma[None, :]
                           qkv_g=(qkv32 * g0[None, :]).to(dtype).contiguous(), wi_g=(ff.wi.weight.float() * g1[None, :]).to(dtype).contiguous()))
        self._rt, self._rt_dtype = rt, dtype
        self._bias_cache.clear()

    def position_bias(self, L: int) -> torch.Tensor:
        if L not in self._bias_cache:
            pos = torch.arange(L, device=self.device_)
            rel = pos[None, :] - pos[:, None]
            nb = self.buckets // 2
            out = (rel > 0).long() * nb
            a = rel.abs()
            max_exact = nb // 2
            big = max_exact + (torch.log(a.float().clamp(min=1) / max_exact) / math.log(self.max_distance / max_exact) * (nb - max_exact)).long()
            big = torch.minimum(big, torch.full_like(big, nb - 1))
            bucket = out + torch.where(a < max_exact, a, big)
            tab = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight
            self._bias_cache[L] = tab[bucket].permute(2, 0, 1).contiguous().float()
        return self._bias_cache[L]

    T5_STREAM = 64      # dropout stream ids of the text encoder: 62 embedding, 63 final, 64 + 4*block + {0 probs, 1 attn out, 2 ff act, 3 ff out}

    @torch.no_grad()
    def encode(self, ids: torch.Tensor, attn_mask: torch.Tensor, drop_seed: Optional[int] = None, drop_p: float = 0.1, dtype=BF16,
               seed_dev: Optional[torch.Tensor] = None, fused: Optional[bool] = None) -> torch.Tensor:
        """ids, attn_mask [U, L] int64 (device) -> last_hidden_state [U*L, 512] bf16.

        ``drop_seed``: the text encoder is frozen (no_grad) but NOT in eval mode in the reference -- the policy's ``self.train()``
        (allenact_dino_transformer.py:193) switches HF T5's dropout 0.1 on too -- so in train mode its six
        dropout sites per block / stack are applied here as well.  One realisation per unique goal and forward pass (the reference
        re-encodes the goal for every (t, b) row and so draws a fresh mask per row)."""
        if self._rt is None or getattr(self, "_rt_dtype", BF16) != dtype:
            self.sync(dtype)
        U, L = ids.shape
        n = U * L
        # seed_dev: recorded / captured passes read the pass seed from device memory (fresh noise per replay), like the fusion layers' sites
        site = (lambda k: ops.Dropout(drop_seed, k, drop_p, seed_dev=seed_dev)) if drop_seed is not None and drop_p > 0 else (lambda k: None)
        x = ops.embed_gather(self.shared.weight, ids.reshape(-1).contiguous(), dtype=dtype)
        ops.dropout_(x, site(62))
        bias = self.position_bias(L)
        kvalid = attn_mask if attn_mask.dtype == torch.uint8 else attn_mask.to(torch.uint8).contiguous()   # uint8 given: no torch op (recorded steps)
        # small passes (an acting step's 64 goals x 12 tokens; an update's unique goals): norm + GEMM in one launch.  ``fused`` given: the caller fixes the form --
        # acting and update passes must share ONE arithmetic: with t5_dropout_per_row the update encodes every (t, b) row (n = R L >> 8192, two-launch form),
        # so the wrapper pins the acting steps to the two-launch form as well
        fused = dtype == BF16 and (n <= 8192 if fused is None else bool(fused))
        for i, (b, rt) in enumerate(zip(self.encoder.block, self._rt)):
            s0 = self.T5_STREAM + 4 * i
            if fused:
                qkv = ops.gemm_nt_rmsa(x, rt["qkv_g"], n, 3 * D, D, 1e-6)
            else:
                nrm, _, _ = ops.norm_fwd(x, b.layer[0].layer_norm.weight, None, 1e-6, n, rms=True, save_stats=False)
                qkv = ops.gemm_nt(nrm, rt["qkv"], n, 3 * D, D)
            ao, _ = ops.attn_fwd(qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, U, L, self.h, 1.0, bias=bias, kvalid=kvalid, save_lse=False,
                                 drop=site(s0))
            x = ops.gemm_nt(ao, rt["o"], n, D, D, residual=x, drop=site(s0 + 1))
            if fused:
                hdn = ops.gemm_nt_rmsa(x, rt["wi_g"], n, 2048, D, 1e-6, act=ops.ACT_RELU, drop=site(s0 + 2))
            else:
                nrm, _, _ = ops.norm_fwd(x, b.layer[1].layer_norm.weight, None, 1e-6, n, rms=True, save_stats=False)
                hdn = ops.gemm_nt(nrm, rt["wi"], n, 2048, D, act=ops.ACT_RELU, drop=site(s0 + 2))
            x = ops.gemm_nt(hdn, rt["wo"], n, D, 2048, residual=x, drop=site(s0 + 3))
        out, _, _ = ops.norm_fwd(x, self.encoder.final_layer_norm.weight, None, 1e-6, n, rms=True, save_stats=False)
        ops.dropout_(out, site(63))
        return out


# ================================================================================================ shared per-call inputs
class Prep:
    """Per-forward inputs shared by the three towers (built once)."""
    pass


class _TowerFn(torch.autograd.Function):
    """One autograd node per tower: explicit kernel schedules inside, reference-style ``loss.backward()`` outside."""

    @staticmethod
    def forward(ctx, anchor, tower, prep, want_logits, want_values):
        need = bool(ctx.needs_input_grad[0])   # anchor requires grad <=> grad mode was on at apply() time
        logits, values, saved = tower.run_forward(prep, need_grad=need)
        ctx.tower, ctx.prep, ctx.saved = tower, prep, saved
        ctx.want = (want_logits, want_values)
        full = tower._last_full_logits
        ctx.has_full = full is not None
        return logits, values, (full if full is not None else logits.new_zeros(0))

    @staticmethod
    def backward(ctx, dlogits, dvalues, dfull):
        wl, wv = ctx.want
        ctx.tower.run_backward(ctx.prep, ctx.saved, dlogits.contiguous() if wl else None, dvalues.contiguous() if wv else None,
                               dfull.contiguous() if (wv and ctx.has_full) else None)
        ctx.saved = None
        return None, None, None, None, None


class SafeDinoLLAMATxNavActorCriticSeparate(Tower):
    """Actor tower (= self) + ``critic_tsfm`` + ``c_critic_tsfm`` (separate_actor_critic.py:8-37)."""

    def __init__(self, device="cuda", tokenizer: Optional[GoalTokenizer] = None, max_steps: int = 500,
                 goal_sensor_uuid="natural_language_spec", rgb_dino_preprocessor_uuid="rgb_dinov2",
                 manipulation_rgb_dino_preprocessor_uuid="manipulation_rgb_dinov2", an_object_is_in_hand_uuid="an_object_is_in_hand",
                 time_step_uuid="time_step", traj_idx_uuid="traj_index", critic_type: str = "linear", precision: str = "bf16", **unused):
        if not torch.cuda.is_available():
            raise RuntimeError("safevla_amd needs an MI355X: there is no CPU or eager fallback for the policy kernels")
        ops.lib()  # fail loudly if the HIP extension is missing
        arena = _Arena()
        device = torch.device(device)
        arena.begin_tower()
        super().__init__(arena, device, max_steps=max_steps, critic_type=critic_type, precision=precision)      # every tower is built from the same kwargs
        arena.begin_tower()
        self.critic_tsfm = Tower(arena, device, max_steps=max_steps, critic_type=critic_type, precision=precision)
        arena.begin_tower()
        self.c_critic_tsfm = Tower(arena, device, max_steps=max_steps, critic_type=critic_type, precision=precision)
        arena.build(device)
        self.towers = [self, self.critic_tsfm, self.c_critic_tsfm]
        for t in self.towers:
            t.bind()
        self.tokenizer = tokenizer or GoalTokenizer()
        self.uuids = dict(goal=goal_sensor_uuid, nav=rgb_dino_preprocessor_uuid, manip=manipulation_rgb_dino_preprocessor_uuid,
                          hand=an_object_is_in_hand_uuid, time=time_step_uuid, traj=traj_idx_uuid)
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        # small shapes (acting steps, small-batch updates): towers on concurrent streams; SVLA_SERIAL_TOWERS=1 turns it off
        import os as _os
        self.concurrent_towers = _os.environ.get("SVLA_SERIAL_TOWERS", "0") != "1"
        self.concurrent_tower_tokens = 1 << 17          # rows x fusion tokens up to which the towers run concurrently
        # Train-mode dropout inside the frozen T5 encoder: by default one realisation per UNIQUE goal per forward (the encoder runs once per
        # unique goal); the reference draws one per (t, b) row (it encodes every row).  True = the reference's statistics, at the price of
        # encoding R rows instead of U unique goals (+13 % FLOPs at L = 12).  Token-id goals only; eval mode is unaffected.
        self._t5_dropout_per_row = False
        # recorded acting steps replay the three towers as grouped launches (ops.GroupedPlans); SVLA_GROUPED_TOWERS=0: the three-stream replay (A/B, tests)
        self.grouped_towers = os.environ.get("SVLA_GROUPED_TOWERS", "1") != "0"
        self._acting_graphs, self._acting_backend = None, "plan"
        if self.concurrent_towers and precision == "bf16" and _os.environ.get("SVLA_NO_ACTING_PLANS", "0") != "1":
            self.enable_acting_plans(True)              # recorded single-step launches are the default acting path
        self._goal_cache: Dict[int, List[int]] = {}
        self.sync_weights()

    # ---- AllenAct ActorCriticModel API bits -------------------------------------------------------------------
    @property
    def t5_dropout_per_row(self) -> bool:
        """one T5 dropout realisation per (t, b) row and tower, as the reference draws it (default: one per unique goal and pass).  Setting it also pins the frozen
        encoder's RMSNorm form: per-row update passes are too large for the norm-fused small-M GEMM, so the acting steps take the two-launch form too -- rollout
        log-probs and the update's recomputed log-probs then come from the same arithmetic."""
        return self._t5_dropout_per_row

    @t5_dropout_per_row.setter
    def t5_dropout_per_row(self, on: bool) -> None:
        on = bool(on)
        if on != self._t5_dropout_per_row:
            self._t5_dropout_per_row = on
            for t in self.towers:
                t.t5_fused = False if on else None
            self.invalidate_recorded()

    @property
    def recurrent_memory_specification(self):
        return None

    def sampler_select(self, keep: list):
        """AllenAct hook: keep only the listed samplers' KV-cache rows (allenact_dino_transformer.py:197-199)."""
        for t in self.towers:
            t.cache_select(keep)

    def trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    # ---- weights ----------------------------------------------------------------------------------------------
    def sync_weights(self, frozen: bool = True):
        """Refresh bf16 mirrors / transposed copies after the fp32 masters changed (load_state_dict, optimiser step)."""
        ar = self.arena
        ops.cast_bf16(ar.flat_p, ar.flat_bf16)
        for t in self.towers:
            t.refresh_transposes()
            if frozen:
                t.visual_encoder.text_encoder.sync(t.adt)
                t._t5_cache = (None, None)
        if frozen:
            # T5Frozen.sync() REPLACES the encoder's runtime tensors: recorded acting steps (and the engine's recorded env-chunks, through
            # the hook below) hold the old ones alive and would keep acting on the previous encoder while updates use the new one
            self.invalidate_recorded()

    def set_fp8_attention(self, on: bool) -> None:
        """BASELINE config 5: run the full-sequence fusion-encoder attention of all three towers (acting and update passes) on the
        e4m3 / e5m2 kernels.  Recorded acting steps are dropped: they hold the other kernels' launches."""
        for t in self.towers:
            t.fp8_attention = bool(on)
        self.invalidate_recorded()

    def invalidate_recorded(self):
        """Drop every recorded launch sequence / captured graph that may reference replaced tensors."""
        if getattr(self, "_acting_graphs", None) is not None:
            self._acting_graphs.clear()
        for hook in getattr(self, "_invalidate_hooks", []):
            hook()

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self.sync_weights()
        return r

    def zero_grad(self, set_to_none: bool = False):
        self.arena.flat_g.zero_()

    # ---- observation preprocessing shared by the towers --------------------------------------------------------
    @torch.no_grad()
    def prepare(self, observations: Dict[str, torch.Tensor], prev_actions: torch.Tensor, masks: torch.Tensor) -> Prep:
        u = self.uuids
        T, B = prev_actions.shape[:2]
        R = T * B
        dev = self.device_
        p = Prep()
        p.T, p.B, p.R = T, B, R
        if "dino_tokens" in observations:       # storage-native layout [T,B,2,84,384] bf16
            p.tokens = observations["dino_tokens"].reshape(R, 2, NPATCH, self.dino_dim).to(self.adt).contiguous()
        else:
            p.tokens = torch.empty(R, 2, NPATCH, self.dino_dim, device=dev, dtype=self.adt)
            ops.feat_to_tokens(observations[u["nav"]].reshape(R, self.dino_dim, NPATCH).contiguous(), p.tokens, 0)
            ops.feat_to_tokens(observations[u["manip"]].reshape(R, self.dino_dim, NPATCH).contiguous(), p.tokens, 1)
        p.prev_actions = prev_actions.reshape(R).contiguous()
        p.masks = masks.reshape(R).to(F32).contiguous()
        p.hand = observations[u["hand"]].reshape(R).contiguous()
        p.time_step = observations[u["time"]].reshape(R).contiguous()
        p.traj_bt = observations[u["traj"]].reshape(T, B).t().contiguous().to(torch.int32)
        # goals: content-hash rows on the GPU, tokenise each unique string once on the host
        per_row = self.t5_dropout_per_row and self.training and "goal_token_ids" in observations
        if "goal_token_ids" in observations and (T == 1 or per_row):
            # single-step (acting) batches: every env is its own goal row -- no de-duplication, hence no host sync (torch.unique), so the
            # host can run ahead of the GPU while it issues the recorded step.  ``t5_dropout_per_row``: the reference's exact train-mode
            # statistics -- it re-encodes the goal of every (t, b) row, so every row sees its own dropout realisation of the frozen T5
            p.ids = observations["goal_token_ids"].reshape(R, -1).contiguous()
            p.attn_mask = (p.ids != 0).to(torch.int64)
            p.attn_mask[:, 0] = 1
            inv = torch.arange(R, device=dev)
        elif "goal_token_ids" in observations:
            ids_rows = observations["goal_token_ids"].reshape(R, -1).contiguous()
            hashes = ops.row_hash(ids_rows.view(torch.uint8).view(R, -1))
            uniq, inv = torch.unique(hashes, return_inverse=True)
            first = torch.full((uniq.numel(),), R, device=dev, dtype=torch.int64).scatter_reduce_(0, inv, torch.arange(R, device=dev), "amin")
            p.ids = ids_rows[first]
            p.attn_mask = (p.ids != 0).to(torch.int64)
            p.attn_mask[:, 0] = 1
        else:
            goal = observations[u["goal"]].reshape(R, -1).contiguous()
            if goal.dtype != torch.uint8:
                goal = goal.to(torch.uint8)
            hashes = ops.row_hash(goal)
            uniq, inv = torch.unique(hashes, return_inverse=True)
            first = torch.full((uniq.numel(),), R, device=dev, dtype=torch.int64).scatter_reduce_(0, inv, torch.arange(R, device=dev), "amin")
            rows = goal[first].cpu().numpy()          # U x 1000 bytes: the only device->host copy of the forward
            keys = uniq.cpu().tolist()
            enc = []
            for k, row in zip(keys, rows):
                if k not in self._goal_cache:
                    self._goal_cache[k] = self.tokenizer.encode(bytes_to_str(row))
                enc.append(self._goal_cache[k])
            L = max(len(e) for e in enc)              # pad to the batch max (reference: padding=True, no fusion mask)
            ids = torch.zeros(len(enc), L, dtype=torch.int64)
            am = torch.zeros(len(enc), L, dtype=torch.int64)
            for i, e in enumerate(enc):
                ids[i, :len(e)] = torch.tensor(e)
                am[i, :len(e)] = 1
            p.ids, p.attn_mask = ids.to(dev), am.to(dev)
            p.ids_key = (tuple(tuple(e) for e in enc), L)      # host-side identity of the goal batch (eval-mode T5 cache)
        p.gid = inv.to(torch.int32).contiguous()
        p.attn_mask_u8 = p.attn_mask.to(torch.uint8).contiguous()      # T5 key-padding mask in the kernels' format (once, not per tower)
        p.U, p.L = p.ids.shape
        p.S = TEXT_OFF + p.L
        return p

    # ---- small shapes: the three (independent) towers on three HIP streams ------------------------------------------------------
    def run_towers_concurrently(self, fn):
        """``[fn(k, tower) for k, tower in enumerate(towers)]`` with tower k issued on its own HIP stream.

        Measured (tools/replay_probe.py): a single-step acting forward is 313 kernels in 4.3 ms whether it is issued eagerly from Python or
        replayed from a pre-bound call list -- it is bound by the GPU-side dispatch of ~100 tiny DEPENDENT kernels per tower (~14 us each),
        not by host issue.  The towers share no intermediate, so running them on three streams overlaps three dependency chains; at
        update-sized shapes (every kernel fills the chip, activations of one tower are tens of GB) the towers stay sequential.
        Inputs were produced on the current stream (side streams wait for it), outputs are handed back to it (it waits for the side
        streams; returned tensors are ``record_stream``-ed so the caching allocator does not recycle them early)."""
        main = torch.cuda.current_stream()
        if getattr(self, "_tower_streams", None) is None:
            self._tower_streams = [torch.cuda.Stream(device=self.device_) for _ in self.towers]
        outs = []
        for k, (t, s) in enumerate(zip(self.towers, self._tower_streams)):
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(fn(k, t))
        for s in self._tower_streams:
            main.wait_stream(s)

        def _rec(o):
            if isinstance(o, torch.Tensor):
                o.record_stream(main)
            elif isinstance(o, (tuple, list)):
                for x in o:
                    _rec(x)
        _rec(outs)
        return outs

    # ---- captured acting step -------------------------------------------------------------------------------------------
    def enable_acting_graphs(self, on: bool = True, backend: str = "hipgraph"):
        """Single-step (acting) forwards are ~100 small dependent kernels per tower.  With this switch the three-tower step is made
        step-independent -- step counter, KV-cache write slot and dropout seeds in device memory, observations copied into static buffers,
        attention over the whole cache window behind the mask (same kernels, same arithmetic) -- and then either
          backend="plan"     recorded once per (envs, goal length) as three ``ops.LaunchPlan``s (one per tower, each on its own HIP stream)
                             and re-issued from a tight loop: the DEFAULT acting path (``enable_acting_plans``), or
          backend="hipgraph" captured as a HIP graph and replayed: measured slower than eager issue on this ROCm stack, kept opt-in."""
        assert backend in ("plan", "hipgraph")
        self._acting_graphs = {} if on else None
        self._acting_backend = backend

    def enable_acting_plans(self, on: bool = True):
        self.enable_acting_graphs(on, backend="plan")

    def _acting_step_graph(self, prep: Prep):
        B, L = prep.B, prep.L
        plan_mode = self._acting_backend == "plan"
        for t in self.towers:
            t._ensure_caches(B)
        key = (B, L, self.training, self._acting_backend, tuple(getattr(t, "_kv_version", 0) for t in self.towers))
        st = self._acting_graphs.get(key)
        dev = self.device_
        if st is None:
            st = Prep()
            st.T, st.B, st.R, st.U, st.L, st.S = 1, B, B, B, L, TEXT_OFF + L
            st.tokens = torch.zeros(B, 2, NPATCH, self.dino_dim, device=dev, dtype=self.adt)
            st.prev_actions = torch.zeros(B, device=dev, dtype=torch.int64)
            st.masks = torch.zeros(B, device=dev, dtype=F32)
            st.hand = torch.zeros(B, device=dev, dtype=torch.int64)
            st.time_step = torch.zeros(B, device=dev, dtype=torch.int64)
            st.traj_bt = torch.zeros(B, 1, device=dev, dtype=torch.int32)
            st.ids = torch.zeros(B, L, device=dev, dtype=torch.int64)
            st.attn_mask = torch.ones(B, L, device=dev, dtype=torch.int64)
            st.gid = torch.arange(B, device=dev, dtype=torch.int32)
            st.t_dev = torch.zeros((), device=dev, dtype=torch.int64)
            st.attn_mask_u8 = torch.ones(B, L, device=dev, dtype=torch.uint8)
            st.kvalid_static = torch.zeros(B, self.max_steps, device=dev, dtype=torch.uint8) if plan_mode else None
            st.ar_steps = torch.arange(self.max_steps, device=dev)
            st.graph = st.plans = None
            for k, t in enumerate(self.towers):
                t._ensure_caches(B)
                t._ar_steps = torch.arange(t.max_steps, device=dev)
                if getattr(t, "_seed_dev_buf", None) is None:
                    t._seed_dev_buf = torch.tensor([(t.drop_seed_base * 0x9E3779B1) & 0x7FFFFFFF], device=dev, dtype=torch.int32)
            for k_old in [k_ for k_ in self._acting_graphs if k_[4] != key[4]]:      # plans of replaced KV caches keep B x max_steps x 1024 x layers x 3 alive
                del self._acting_graphs[k_old]
            self._acting_graphs[key] = st
        for t in self.towers:
            t.refresh_folded()          # (recorded / captured steps do not run the Python that would notice an optimiser step: refresh the gamma-folded weights here, in place)
        # per-step inputs -> static buffers
        st.tokens.copy_(prep.tokens); st.prev_actions.copy_(prep.prev_actions); st.masks.copy_(prep.masks); st.hand.copy_(prep.hand)
        st.time_step.copy_(prep.time_step)
        st.ids.copy_(prep.ids[prep.gid.long()]); st.attn_mask.copy_(prep.attn_mask[prep.gid.long()])
        st.attn_mask_u8.copy_(st.attn_mask)
        st.t_dev.fill_(self.time_step_counter)
        if plan_mode:
            # recorded step: every tower attends to cache slots [max(t - time_step_b, 0), t] (allenact_dino_transformer.py:388-397), computed once
            tc, ar = self.time_step_counter, st.ar_steps
            st.kvalid_static.copy_((ar[None, :] <= tc) & (ar[None, :] >= torch.clamp(tc - st.time_step, min=0)[:, None]))
            if st.plans is None:
                for t in self.towers:
                    t._ensure_caches(B)
                versions = tuple(getattr(t, "_kv_version", 0) for t in self.towers)

                def rec(k, t):
                    plan = ops.LaunchPlan()
                    t._t_dev, t._seed_dev = st.t_dev, t._seed_dev_buf
                    keep = t.time_step_counter
                    with plan:
                        lg, vl, _ = t.run_forward(st, need_grad=False)
                    t.time_step_counter = keep
                    t._t_dev, t._seed_dev = None, None
                    return plan, lg, vl
                for t in self.towers:
                    t._seed_dev_buf.add_(0x3C6EF35)
                res = self.run_towers_concurrently(rec)          # the recording pass is a real step
                assert versions == tuple(getattr(t, "_kv_version", 0) for t in self.towers)
                st.plans, st.outs = [r[0] for r in res], [(r[1], r[2]) for r in res]
                st.gplans = {}
                st.grouped_ok = ops.GroupedPlans.compatible(st.plans)
            elif self.grouped_towers and st.grouped_ok:
                # tower-grouped replay (round 6): call i of the three recorded sequences is issued as ONE grid whose blockIdx.z picks the tower's
                # arguments (csrc/launch.h) -- one dependency chain on the current stream instead of three chains on three streams, three times
                # the workgroups per dispatch.  Same kernels, same arithmetic: bit-identical to the three-stream replay (tests/test_grouped_gpu.py)
                for t in self.towers:
                    t._seed_dev_buf.add_(0x3C6EF35)              # fresh dropout noise per step (device-resident seed, wraps in int32)
                self._grouped_replay(st)
            else:
                main = torch.cuda.current_stream()
                for t in self.towers:
                    t._seed_dev_buf.add_(0x3C6EF35)              # fresh dropout noise per step (device-resident seed, wraps in int32)
                # one foreign call per tower (svla_replay_calls).  Host issue 2.0 -> 1.0 ms per step; the step itself is bound by the GPU side
                # (~310 small dependent kernels on three streams: 3.0 ms), so issuing the three sequences from three host threads -- tried,
                # ctypes drops the GIL during the call -- changes nothing (21.4 k env-steps/s either way)
                for plan, s_ in zip(st.plans, self._tower_streams):
                    s_.wait_stream(main)
                    plan.replay()
                for s_ in self._tower_streams:
                    main.wait_stream(s_)
            for t in self.towers:
                t.time_step_counter += 1
            return st.outs[0][0].clone(), st.outs[1][1].clone(), st.outs[2][1].clone()

        def body():
            outs = []
            for t in self.towers:
                t._t_dev, t._seed_dev = st.t_dev, t._seed_dev_buf
                t._seed_dev_buf.add_(0x3C6EF35)          # fresh dropout noise per replay (wraps in int32)
                keep = t.time_step_counter
                lg, vl, _ = t.run_forward(st, need_grad=False)
                t.time_step_counter = keep               # the host counter is advanced once per step below
                t._t_dev, t._seed_dev = None, None
                outs.append((lg, vl))
            return outs

        if st.graph is None:
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):                  # warm-up outside capture (lazy initialisations, allocator)
                body()
            torch.cuda.current_stream().wait_stream(s_)
            st.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st.graph):
                st.outs = body()
        st.graph.replay()
        for t in self.towers:
            t.time_step_counter += 1
        return st.outs[0][0].clone(), st.outs[1][1].clone(), st.outs[2][1].clone()

    def _grouped_replay(self, st):
        """the recorded step of the three towers as grouped launches on the current stream.  (Issuing the text path -- frozen T5 + text adapter, ~40 small launches that
        leave most CUs idle -- on a side stream next to the visual compressor's GEMMs was measured: 2.324 vs 2.321 ms per step, not kept.)"""
        main = torch.cuda.current_stream()
        gp = st.gplans.get(main.cuda_stream)
        if gp is None:
            gp = st.gplans[main.cuda_stream] = ops.GroupedPlans(st.plans, main.cuda_stream)
        gp.replay()

    def _acting_fast(self, observations, prev_actions, masks):
        """A recorded, tower-grouped step issued straight from the observation dict: ONE staging launch (svla_acting_stage: inputs -> static buffers, T5 padding masks,
        KV-window mask, step counter, seed bumps) + the grouped replay -- no ``prepare``, none of the ~30 framework copies / compares the general path spends on a
        step.  Returns None whenever anything is not exactly as the recorded step expects (first step of a shape, other dtypes / layouts, string goals, a cache
        window about to wrap): the general path then handles -- and, if needed, records -- the step."""
        if getattr(self, "_acting_graphs", None) is None or self._acting_backend != "plan" or not self.grouped_towers or torch.is_grad_enabled():
            return None
        tk, ids = observations.get("dino_tokens"), observations.get("goal_token_ids")
        if tk is None or ids is None or prev_actions.dim() != 2 or prev_actions.shape[0] != 1:
            return None
        B, L, tc = prev_actions.shape[1], ids.shape[-1], self.time_step_counter
        if not (tc < self.max_steps - 1 and all(t.time_step_counter == tc and getattr(t, "_kv", None) is not None and t._kv[0].shape[0] >= B for t in self.towers)):
            return None
        st = self._acting_graphs.get((B, L, self.training, self._acting_backend, tuple(getattr(t, "_kv_version", 0) for t in self.towers)))
        if st is None or st.plans is None or not getattr(st, "grouped_ok", False):
            return None
        u = self.uuids
        hand, ts = observations.get(u["hand"]), observations.get(u["time"])
        i64 = torch.int64
        if hand is None or ts is None or not (tk.dtype == self.adt and tk.is_contiguous() and tk.numel() == st.tokens.numel() and prev_actions.dtype == i64 and prev_actions.is_contiguous()
                                              and masks.dtype == F32 and masks.is_contiguous() and masks.numel() == B and hand.dtype == i64 and hand.is_contiguous() and hand.numel() == B
                                              and ts.dtype == i64 and ts.is_contiguous() and ts.numel() == B and ids.dtype == i64 and ids.is_contiguous() and ids.numel() == B * L):
            return None
        for t in self.towers:
            t.refresh_folded()
        ops.acting_stage(tk, st.tokens, prev_actions, st.prev_actions, masks, st.masks, hand, st.hand, ts, st.time_step, ids, st.ids, st.attn_mask, st.attn_mask_u8,
                         st.kvalid_static, st.t_dev, B, L, self.max_steps, tc, [t._seed_dev_buf for t in self.towers], 0x3C6EF35)
        self._grouped_replay(st)
        for t in self.towers:
            t.time_step_counter += 1
        return SafeActorCriticOutput(distributions=CategoricalDistr(st.outs[0][0].clone()), values=st.outs[1][1].clone(), c_values=st.outs[2][1].clone(), extras={})

    # ---- reference forward API -------------------------------------------------------------------------------------
    def forward(self, observations, memory, prev_actions, masks):
        fast = self._acting_fast(observations, prev_actions, masks)
        if fast is not None:
            return fast, memory
        prep = self.prepare(observations, prev_actions, masks)
        if (prep.T == 1 and getattr(self, "_acting_graphs", None) is not None and not torch.is_grad_enabled()
                and self.time_step_counter < self.max_steps - 1 and all(t.time_step_counter == self.time_step_counter for t in self.towers)):
            logits, values, c_values = self._acting_step_graph(prep)
            return SafeActorCriticOutput(distributions=CategoricalDistr(logits), values=values, c_values=c_values, extras={}), memory
        if prep.T == 1 and torch.is_grad_enabled():
            raise RuntimeError("single-step (acting) forwards run under torch.no_grad(), as in the reference's rollout collection")
        if not torch.is_grad_enabled() and self.concurrent_towers and prep.R * prep.S <= self.concurrent_tower_tokens:
            (logits, _), (_, values), (_, c_values) = self.run_towers_concurrently(lambda k, t: t.run_forward(prep, need_grad=False)[:2])
            c_full = self.c_critic_tsfm._last_full_logits
            extras = self._extras(c_values, c_full) if prep.T > 1 else {}
            return SafeActorCriticOutput(distributions=CategoricalDistr(logits), values=values, c_values=c_values, extras=extras), memory
        logits, _, _ = _TowerFn.apply(self._anchor, self, prep, True, False)
        _, values, _ = _TowerFn.apply(self._anchor, self.critic_tsfm, prep, False, True)
        _, c_values, c_full = _TowerFn.apply(self._anchor, self.c_critic_tsfm, prep, False, True)
        # the diagnostics in ``extras`` are consumed by the losses only (update batches); single-step acting forwards skip them
        extras = self._extras(c_values, c_full) if prep.T > 1 else {}
        return SafeActorCriticOutput(distributions=CategoricalDistr(logits), values=values, c_values=c_values, extras=extras), memory

    def _extras(self, c_values, c_full=None):
        """``extras`` of the COST-critic tower, which is what the 3-tower wrapper hands on (separate_actor_critic.py:31-36); keys per
        allenact_dino_transformer.py:425-463: total_norm (the tower's gradient norm as left by the previous backward), stop_grad_values
        (or stop_grad_logits / full_logits / loss_func for the discrete critic), weight_norm / bias_norm / weight_grad_norm of the last
        critic layer.  Device tensors (the reference builds host tensors with one sync each)."""
        tw = self.c_critic_tsfm
        with torch.no_grad():
            fc = tw.critic.fc if tw.critic_type == "linear" else tw.critic.fc[-1]
            a, b = self.arena.tower_ranges[2]
            sq = torch.zeros(1, device=self.device_, dtype=torch.float64)
            ops.sumsq(self.arena.flat_g[a:b], sq)
            extras = {"total_norm": sq.sqrt().float(), "weight_norm": fc.weight.norm(2).reshape(1), "bias_norm": fc.bias.norm(2).reshape(1),
                      "weight_grad_norm": tw.g(fc.weight).norm(2).reshape(1)}
        if tw.critic_type == "discrete" and c_full is not None and c_full.numel():
            extras.update(full_logits=c_full, stop_grad_logits=c_full.detach(), loss_func=tw.critic.loss_fn)
        else:
            extras["stop_grad_values"] = c_values.detach()
        return extras
