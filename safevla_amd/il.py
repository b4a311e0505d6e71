"""Offline imitation-learning workload on the same kernels.

Mirrors ``EarlyFusionCnnTransformer`` with the llama decoder, in every ``model_version`` the reference can construct (512 x 8 and 768 x 12 on the MFMA attention kernels; 768 x 8 = heads of 96 on the fp32 ones)
(/root/reference/architecture/models/transformer_models/early_fusion_tsfm_models.py:49-207,221-312; ``VERSIONS`` below): text-conditioned multi-camera
encoder (text_cond_visual_encoder.py:56-268) -> + last-action / in-hand / time embeddings (:120-157) -> causal llama decoder ->
``actor`` -> ``nn.CrossEntropyLoss(ignore_index=-1)`` (:93,115-117); and the optimiser of ``training/offline/train_pl.py:283-287``
(AdamW, lr 1e-4).  It IS one tower of the RL model -- the RL towers are initialised from exactly these weights
(``checkpoint.init_towers_from_il``) -- so the forward/backward run the update path's kernel schedules (``model.Tower``); new here:
the batch-first IL batch format, the fused cross-entropy kernel and decoupled weight decay in the Adam kernel.

The frozen image encoder stays outside, as on the RL path: visual sensors are either pre-encoded features ``[B,T,C,7,12]`` (C = 384 DINOv2-S,
768 DINOv2-B / SigLIP-B, 1024 SigLIP-L, 2048 CLIP RN50) or uint8 frames ``[B,T,H,W,3]`` (then the frozen ViT of ``preproc`` runs first: DINOv2 on
224 x 384, SigLIP on 256 x 256 frames).  The frozen text encoder is t5-small or, for the ``siglip_*`` presets, the SigLIP text tower
(``siglip_text.SigLIPTextFrozen``: ``goals`` is then the tokenizer's id tensor [B, 64], preprocessors.py:334-343).  ``state_dict`` uses the reference's names
(``actor.weight``, no critic head), so Lightning checkpoints (``model.`` prefix) interchange.
"""
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from .model import BF16, DINO, F32, N_ACTIONS, NPATCH, TEXT_OFF, Prep, Tower, _Arena, _TowerFn

NAV, MANIP = "raw_navigation_camera", "raw_manipulation_camera"
START_TOKEN, PAD_TOKEN = N_ACTIONS, N_ACTIONS + 1      # last_actions vocabulary (:95-101)


class _CEFn(torch.autograd.Function):
    """Fused cross-entropy forward+backward (svla_ce_loss_fwd_bwd_f32): mean over targets != ignore_index."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        rows, A = logits.shape
        n_valid = (target != ignore_index).sum().to(F32).reshape(1)
        dlogits = torch.empty_like(logits)
        sums = torch.zeros(1, device=logits.device, dtype=torch.float64)
        ops.ce_loss_fwd_bwd(logits.contiguous(), target.contiguous(), n_valid, dlogits, sums, ignore_index)
        ctx.save_for_backward(dlogits)
        return (sums[0] / n_valid[0].clamp(min=1.0).double()).to(F32)

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None, None


class EarlyFusionCnnTransformer(Tower):
    # model_version -> (fusion layers, decoder layers, image-feature width, text encoder); early_fusion_tsfm_models.py:221-312.  Every preset whose
    # fusion transformer and decoder are TransformerConfig(n, 512, 8), with the llama decoder (``use_llama_decoder`` defaults to True, :46):
    # DINOv2-S / -B, SigLIP-B / -L (image trunk + text tower) and CLIP RN50 (pre-encoded features only: its conv trunk is not built).
    # and the 768-wide presets whose heads are 64 wide (12 heads: siglip_base_6_3 / _6_6 / _12_12; same kernels at D = 768, llama hidden 2048).
    # TransformerConfig(n, 768, 8) = heads of 96 (base_6, and the fusion transformer of siglip_base_3_6) builds too, with its attention on the fp32 kernels.
    # Four more names cannot be constructed in the reference as shipped: small_3_nonTxEnc / siglip_base_3_nonTxEnc (``globals()["NonTxMultiCameraVisualEncoder"]``, :64, is a KeyError: the class is not
    # imported into that module, :22-28) and siglip_base_384_3 / siglip_base_384_resize_3 (image encoders absent from IMAGE_ENCODERS, image_encoders.py:103-112).
    VERSIONS = {"small": (3, 3, 384, "t5-small"), "small_3": (3, 3, 384, "t5-small"), "small_6": (6, 6, 384, "t5-small"), "base_3": (3, 3, 768, "t5-small"),
                "siglip_base_3": (3, 3, 768, "SigLIPBase"), "siglip_3": (3, 3, 768, "SigLIPBase"), "siglip_base_3_llama": (3, 3, 768, "SigLIPBase"),
                "siglip_base_6": (6, 6, 768, "SigLIPBase"), "siglip_large_3": (3, 3, 1024, "SigLIPLarge"), "clip_resnet_50_3": (3, 3, 2048, "t5-small"),
                "siglip_base_6_3": (6, 3, 768, "SigLIPBase", 768, 12), "siglip_base_6_6": (6, 6, 768, "SigLIPBase", 768, 12),
                "siglip_base_12_12": (12, 12, 768, "SigLIPBase", 768, 12),
                # heads of 96 (768 / 8): attention on the fp32 kernels (ops.attn_fwd head_dim != 64), everything else as above -- a slow path
                "base_6": (6, 6, 768, "t5-small", 768, 8), "siglip_base_3_6": (3, 6, 768, "SigLIPBase", 768, 8, 12)}

    def __init__(self, device="cuda", max_length: int = 1000, input_sensors=(NAV, MANIP, "last_actions", "an_object_is_in_hand"),
                 image_preprocessor=None, n_fusion_layers: int = 3, n_decoder_layers: int = 3, dino_dim: int = DINO, text_encoder: str = "t5-small", d_model: int = 512, n_heads: int = 8,
                 n_heads_decoder: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("safevla_amd needs an MI355X: there is no CPU or eager fallback for the policy kernels")
        ops.lib()
        arena = _Arena()
        device = torch.device(device)
        super().__init__(arena, device, n_fusion_layers=n_fusion_layers, n_decoder_layers=n_decoder_layers, max_steps=max_length, dino_dim=dino_dim,
                         text_encoder=text_encoder, d_model=d_model, n_heads=n_heads, n_heads_decoder=n_heads_decoder)
        arena.build(device)
        self.bind()
        self.towers = [self]
        self.input_sensors = list(input_sensors)
        self.image_preprocessor = image_preprocessor
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self.sync_weights()

    # ---- weights: reference names ------------------------------------------------------------------------------------------
    @staticmethod
    def _to_ref(k: str) -> Optional[str]:
        if k.startswith("critic."):
            return None
        return k.replace("actor.linear.", "actor.")

    def state_dict(self, *a, **kw):
        sd = super().state_dict(*a, **kw)
        return type(sd)((self._to_ref(k), v) for k, v in sd.items() if self._to_ref(k) is not None)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        own = super().state_dict()
        sd = {k.replace("actor.weight", "actor.linear.weight").replace("actor.bias", "actor.linear.bias"): v for k, v in state_dict.items()}
        missing = [k for k in own if k not in sd and not k.startswith("critic.")]
        unexpected = [k for k in sd if k not in own and "visual_encoder.image_encoder" not in k]
        if strict and (missing or unexpected):
            raise RuntimeError(f"state_dict mismatch: missing {missing[:5]}, unexpected {unexpected[:5]}")
        with torch.no_grad():
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v.to(own[k].device, own[k].dtype))
        self.sync_weights()
        return missing, unexpected

    def sync_weights(self):
        ar = self.arena
        ops.cast_bf16(ar.flat_p, ar.flat_bf16)
        self.refresh_transposes()
        self.visual_encoder.text_encoder.sync()
        self._t5_cache = (None, None)

    def zero_grad(self, set_to_none: bool = False):
        self.arena.flat_g.zero_()

    def trainable_parameters(self):
        return [p for n, p in self.named_parameters() if p.requires_grad and not n.startswith("critic.")]

    # ---- batch -> kernel inputs ----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def prepare(self, batch: Dict) -> Prep:
        dev = self.device_
        nav = batch[NAV]
        B, T = nav.shape[:2]
        R = T * B
        p = Prep()
        p.T, p.B, p.R = T, B, R
        p.acting = False                                         # forward(batch): whole windows, also when T = 1 (the agent's single steps set it)
        p.tokens = torch.empty(R, 2, NPATCH, self.dino_dim, device=dev, dtype=BF16)
        for cam, key in enumerate((NAV, MANIP)):
            x = batch[key].to(dev)
            if x.dtype == torch.uint8:                           # raw frames [B,T,H,W,3] -> frozen ViT
                if self.image_preprocessor is None:
                    self.image_preprocessor = self._frozen_image_encoder(key, dev)
                fr = x.transpose(0, 1).reshape(R, *x.shape[2:]).contiguous()
                self.image_preprocessor.process_tokens(fr, p.tokens, cam)
            else:                                                # pre-encoded features [B,T,384,7,12]
                ops.feat_to_tokens(x.transpose(0, 1).reshape(R, self.dino_dim, NPATCH).contiguous().float(), p.tokens, cam)
        tb = lambda v: v.to(dev).transpose(0, 1).reshape(R).contiguous()     # [B,T] -> rows (t*B + b)
        la = tb(batch["last_actions"]).to(torch.int64)
        p.prev_actions = la
        p.masks = (la != START_TOKEN).to(F32)                    # start token <=> "no previous action" (mask 0 selects row 20)
        p.hand = tb(batch["an_object_is_in_hand"]).to(torch.int64) if "an_object_is_in_hand" in batch else torch.zeros(R, device=dev, dtype=torch.int64)
        p.time_step = tb(batch["time_ids"]).to(torch.int64)
        p.traj_bt = torch.arange(B, device=dev, dtype=torch.int32)[:, None].expand(B, T).contiguous()   # one trajectory per row: causal
        goals = batch["goals"]
        if self.text_encoder_name == "t5-small":                 # HF tokenizer output (preprocessors.py:157-163)
            p.ids = goals["input_ids"].to(dev).to(torch.int64).contiguous()
            p.attn_mask = goals["attention_mask"].to(dev).to(torch.int64).contiguous()
            p.U, p.L = p.ids.shape
        else:                                                    # open_clip tokenizer output: ids [B, context] (preprocessors.py:334-343); + the pooled token
            p.ids = (goals["input_ids"] if isinstance(goals, dict) else goals).to(dev).to(torch.int64).contiguous()
            p.attn_mask = torch.ones_like(p.ids)
            p.U, p.L = p.ids.shape[0], self.visual_encoder.text_encoder.out_tokens(p.ids.shape[1])
        p.gid = (torch.arange(R, device=dev) % B).to(torch.int32).contiguous()
        p.S = TEXT_OFF + p.L
        return p

    def _frozen_image_encoder(self, key, dev):
        """IMAGE_ENCODERS of image_encoders.py:103-112 for raw uint8 frames, by preset: DINOv2 (224 x 384) or the SigLIP trunk (256 x 256)."""
        from .preproc import DinoViTPreprocessor, SigLIPPreprocessor
        if self.text_encoder_name.startswith("SigLIP"):
            return SigLIPPreprocessor(key, key, siglip_model_type={768: "ViT-B-16-SigLIP-256", 1024: "ViT-L-16-SigLIP-256"}[self.dino_dim], device=dev)
        if self.dino_dim == 2048:
            raise NotImplementedError("clip_resnet_50_3: the CLIP RN50 conv trunk is not built; pass its pre-encoded (2048, 7, 12) features")
        return DinoViTPreprocessor(key, key, dino_model_type={384: "dinov2_vits14", 768: "dinov2_vitb14", 1024: "dinov2_vitl14"}[self.dino_dim], device=dev)

    # ---- reference forward API --------------------------------------------------------------------------------------------------
    def forward(self, batch: Dict) -> Dict[str, torch.Tensor]:
        prep = self.prepare(batch)
        logits, _, _ = _TowerFn.apply(self._anchor, self, prep, True, False)      # [T, B, A] fp32
        out = dict(actions_logits=logits.transpose(0, 1))
        if "actions" in batch:
            tgt = batch["actions"].to(self.device_).transpose(0, 1).reshape(-1).to(torch.int64)
            loss = _CEFn.apply(logits.reshape(-1, N_ACTIONS), tgt, -1)
            out["actions_loss"] = loss
            out["loss"] = loss
        return out

    @classmethod
    def build_agent(cls, model_version="small_3", input_sensors=(NAV, MANIP, "last_actions", "an_object_is_in_hand"), loss="action", device="cuda",
                    sampling="greedy", ckpt_pth: Optional[str] = None, **kw):
        """``build_agent`` of early_fusion_tsfm_models.py:352-363: the model of ``build_model`` wrapped in its online agent."""
        return EarlyFusionCnnTransformerAgent(cls.build_model(model_version, input_sensors, loss, device=device, ckpt_pth=ckpt_pth), device, sampling, **kw)

    @classmethod
    def version_config(cls, model_version):
        """(fusion layers, decoder layers, image-feature width, text encoder, transformer width, fusion heads, decoder heads) of a preset"""
        v = cls.VERSIONS[model_version]
        v = v + (512, 8)[len(v) - 4:] if len(v) < 6 else v
        return v if len(v) == 7 else v + (v[5],)

    @classmethod
    def build_model(cls, model_version="small_3", input_sensors=(NAV, MANIP, "last_actions", "an_object_is_in_hand"), loss="action",
                    device="cuda", ckpt_pth: Optional[str] = None, ckpt_prefix: str = "model."):
        if model_version not in cls.VERSIONS:
            raise NotImplementedError(f"model_version {model_version!r}: built are {sorted(cls.VERSIONS)} (512-wide fusion transformer + llama decoder; "
                                      "early_fusion_tsfm_models.py:221-312)")
        nf, nd, dd, te, dm, nh, nhd = cls.version_config(model_version)
        m = cls(device=device, input_sensors=input_sensors, n_fusion_layers=nf, n_decoder_layers=nd, dino_dim=dd, text_encoder=te, d_model=dm, n_heads=nh,
                n_heads_decoder=nhd)
        if ckpt_pth is not None:    # Lightning checkpoint (training/offline/train_utils.py:6-68)
            sd = torch.load(ckpt_pth, map_location="cpu")["state_dict"]
            m.load_state_dict({k[len(ckpt_prefix):]: v for k, v in sd.items() if k.startswith(ckpt_prefix)}, strict=False)
        return m


class ILTrainer:
    """``LitModel.training_step`` + ``configure_optimizers`` (training/offline/train_pl.py:154-186,283-287): AdamW(lr) on the flat arena."""

    def __init__(self, model: EarlyFusionCnnTransformer, lr: float = 1e-4, weight_decay: float = 0.01, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model, self.lr, self.wd, self.betas, self.eps = model, lr, weight_decay, betas, eps
        self.step_count = 0

    def training_step(self, batch: Dict) -> Dict[str, float]:
        m = self.model
        m.zero_grad()
        out = m(batch)
        out["loss"].backward()
        self.step_count += 1
        ar = m.arena
        ops.adam_step(ar.flat_p, ar.flat_g, ar.flat_m, ar.flat_v, ar.flat_bf16, self.lr, self.step_count, self.betas[0], self.betas[1],
                      self.eps, weight_decay=self.wd)
        m.refresh_transposes()
        return {"loss": float(out["loss"].detach())}

    def state_dict(self):
        ar = self.model.arena
        return dict(step=self.step_count, exp_avg=ar.flat_m.clone(), exp_avg_sq=ar.flat_v.clone())


class EarlyFusionCnnTransformerAgent:
    """Online agent of the imitation-learning model: ``EarlyFusionCnnTransformerAgent`` of early_fusion_tsfm_models.py:366-520 behind ``AbstractAgent``
    (architecture/agent.py:5-51) -- ``reset()``, ``get_action_list()``, ``get_action(observations, goal_spec) -> (action_str, action_probs)``.

    One call = the step's sensors (uint8 frames -> the preset's frozen image trunk, or pre-encoded features) + the cached goal -> a single-step forward of the tower
    against the llama KV caches (the acting path of ``model.Tower``: the reference re-embeds the step, appends it to ``cache["embedded_features"]`` and calls the
    decoder with ``start_pos = curr_t``, :476-499 -- the same per-step arithmetic) -> greedy or sampled action; the previous action fed to the next step is the
    chosen action index (:512-513), the first step's is the start token (:416-420).

    ``goal_spec``: a string (tokenised by ``tokenizer``: for the t5 presets ``text.GoalTokenizer`` -- the real ``t5-small`` vocabulary through its ``spiece_model`` --,
    for the SigLIP presets any callable ``tokenizer([text], context_length=64) -> ids [1, 64]`` like open_clip's; neither vocabulary ships offline) or already
    tokenised ids (``{"input_ids", "attention_mask"}`` / an id tensor).  ``max_seq_len`` = the KV-cache window: 512 here (the single-query attention kernel's
    limit; the reference's tasks end at 500 steps, training/online/base.py:129) against the reference's 1000; an episode that outlives it restarts the window (the
    reference slides it)."""

    def __init__(self, model: EarlyFusionCnnTransformer, device="cuda", sampling: str = "greedy", max_seq_len: int = 512, tokenizer=None,
                 generator: Optional[torch.Generator] = None):
        from .agent import ALL_STRETCH_ACTIONS
        if sampling not in ("greedy", "sample"):
            raise NotImplementedError(f"sampling {sampling!r}: 'greedy' (argmax) or 'sample' (categorical), utils/nn_utils.py sample_action_index_from_logits")
        if getattr(model, "hdim_dec", 64) != 64:
            # the step-by-step agent runs the KV-cached decoder step, which exists for 64-wide heads only (MFMA / decode attention kernels); say so at
            # construction, not at the first get_action.  forward(batch) -- training and offline evaluation -- works for these presets.
            raise NotImplementedError(f"the online agent needs 64-wide decoder heads; this preset has heads of {model.hdim_dec} "
                                      "(TransformerConfig(n, 768, 8): base_6, siglip_base_3_6) -- use forward(batch) on whole windows")
        self.model, self.device, self.sampling, self.max_seq_len = model, torch.device(device), sampling, min(max_seq_len, 512, model.max_steps)
        self._warned_window = False
        self.action_list = list(ALL_STRETCH_ACTIONS)
        self.generator = generator
        if tokenizer is None and model.text_encoder_name == "t5-small":
            from .text import GoalTokenizer
            tokenizer = GoalTokenizer()
        self.tokenizer = tokenizer
        self.model.eval()
        self.reset()

    def reset(self):
        self.curr_t = 0
        self.cache = dict()
        self._warned_window = False
        self.model.time_step_counter = 0          # KV-cache slot of the next step; the caches themselves are overwritten slot by slot

    def get_action_list(self):
        return self.action_list

    def _goal(self, goal_spec):
        if isinstance(goal_spec, str):
            if self.tokenizer is None:
                raise ValueError("a string goal needs a tokenizer (SigLIP presets: open_clip's tokenizer is not available offline; pass ids or tokenizer=)")
            if self.model.text_encoder_name == "t5-small":
                enc = self.tokenizer([goal_spec], return_tensors="pt")
                return {k: v.to(self.device) for k, v in enc.items()}
            return torch.as_tensor(self.tokenizer([goal_spec], context_length=64)).to(self.device)
        if isinstance(goal_spec, dict):
            return {k: torch.as_tensor(v).to(self.device).reshape(1, -1) for k, v in goal_spec.items()}
        return torch.as_tensor(goal_spec).to(self.device).reshape(1, -1)

    @torch.no_grad()
    def get_action(self, observations: Dict, goal_spec):
        m, dev = self.model, self.device
        if self.curr_t == 0:
            self.cache["goal"] = self._goal(goal_spec)
            g = self.cache["goal"]
            ids = g["input_ids"] if isinstance(g, dict) else g
            self.cache["goal_key"] = tuple(int(v) for v in ids.reshape(-1).tolist())
        slot = self.curr_t % self.max_seq_len
        if slot == 0:
            m.time_step_counter = 0
            if self.curr_t > 0 and not self._warned_window:
                # the reference keeps every embedded step and attends to the last max_seq_len of them (a sliding window, :493-495); here the KV cache restarts:
                # the steps right after a restart see a shorter history than the reference's.  time_ids keeps counting like the reference's (:480).
                import warnings
                warnings.warn(f"episode longer than the {self.max_seq_len}-step KV-cache window: the window restarts (the reference slides it)")
                self._warned_window = True
        one = lambda v, dt: torch.as_tensor(np.asarray(v)).to(dev).to(dt).reshape(1, 1)
        batch = {"goals": self.cache["goal"], "time_ids": one(self.curr_t, torch.int64),      # = curr_t, as early_fusion_tsfm_models.py:480 (not the cache slot)
                 "last_actions": one(START_TOKEN if self.curr_t == 0 else self.cache["last_actions"], torch.int64),
                 "an_object_is_in_hand": one(np.asarray(observations.get("an_object_is_in_hand", 0)).reshape(-1)[0], torch.int64)}
        for key in (NAV, MANIP):
            x = torch.as_tensor(np.ascontiguousarray(observations[key])).to(dev)
            batch[key] = x.reshape((1, 1) + tuple(x.shape))
        prep = m.prepare(batch)
        prep.acting = True                             # a single step against the llama KV caches
        prep.ids_key = self.cache["goal_key"]          # eval mode: the frozen text encoder runs once per episode (the reference caches text_feats at t = 0, :459-465)
        logits, _, _ = m.run_forward(prep, need_grad=False)
        curr = logits.reshape(-1).float()
        probs = torch.softmax(curr, -1)
        idx = int(torch.argmax(curr)) if self.sampling == "greedy" else int(torch.multinomial(probs, 1, generator=self.generator))
        self.cache["last_actions"] = idx
        self.curr_t += 1
        return self.action_list[idx], probs
