"""Checkpoint interchange with the reference's formats.

  * IL -> RL hand-off: Lightning ``state_dict`` with ``model.`` prefix, ``actor.weight/bias -> actor.linear.weight/bias``,
    loaded per tower (/root/reference/training/offline/train_utils.py:6-68; called for every tower at
    architecture/models/allenact_transformer_models/allenact_dino_transformer.py:169-177, separate_actor_critic.py:11,25).
  * RL -> RL: AllenAct-style ``{"model_state_dict": ...}`` (.pt), dropping ``critic_tsfm`` keys on request
    (allenact_dino_transformer.py:178-191; key read at inference_agent.py:142-143).
"""
from typing import Dict, Optional

import torch


def load_pl_ckpt_allenact(model, ckpt, ckpt_prefix: str = "model.", verbose: bool = False):
    """Load an imitation-learning (Lightning) checkpoint into ONE tower (or the 3-tower model's actor tower).
    ``ckpt``: path, an already-loaded Lightning checkpoint (dict with a ``state_dict`` entry) or that state dict itself."""
    if isinstance(ckpt, str):
        ckpt = torch.load(ckpt, map_location="cpu")
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt          # whole Lightning checkpoint or its bare state dict
    sd = {k.replace("actor.weight", "actor.linear.weight").replace("actor.bias", "actor.linear.bias"): v for k, v in sd.items()}
    new = model.state_dict()
    loaded = [k for k in new if ckpt_prefix + k in sd]
    for k in loaded:
        new[k] = sd[ckpt_prefix + k]
    model.load_state_dict(new)
    missing = [k for k in new if ckpt_prefix + k not in sd]
    extra = [k[len(ckpt_prefix):] for k in sd if k[len(ckpt_prefix):] not in new and "visual_encoder.image_encoder.model" not in k]
    if verbose:
        print(f"loaded {len(loaded)} tensors; {len(missing)} not in checkpoint; {len(extra)} checkpoint tensors unused")
    return loaded, missing, extra


def init_towers_from_il(model, ckpt, ckpt_prefix: str = "model."):
    """Every tower starts from the IL weights (encoder, decoder, actor); critic heads stay fresh."""
    out = []
    for tower in model.towers:
        # a tower's own state_dict (without the sibling towers' prefixes)
        own = _TowerView(tower)
        out.append(load_pl_ckpt_allenact(own, ckpt, ckpt_prefix))
    model.sync_weights()
    return out


class _TowerView:
    """state_dict()/load_state_dict() restricted to one tower's own tensors (the actor tower is the top-level module)."""

    def __init__(self, tower):
        self.t = tower

    def _own(self, k):
        return not (k.startswith("critic_tsfm.") or k.startswith("c_critic_tsfm."))

    def state_dict(self):
        return {k: v for k, v in self.t.state_dict().items() if self._own(k)}

    def load_state_dict(self, sd):
        cur = self.t.state_dict()
        with torch.no_grad():
            for k, v in sd.items():
                if self._own(k) and k in cur:
                    cur[k].copy_(v.to(cur[k].device, cur[k].dtype))


def save_checkpoint(path: str, model, engine=None, total_steps: int = 0, extra: Optional[Dict] = None):
    ck = {"model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "total_steps": int(total_steps)}
    if engine is not None:
        ar = model.arena
        ck["optimizer_state"] = {"exp_avg": ar.flat_m.cpu(), "exp_avg_sq": ar.flat_v.cpu(), "step": engine.opt_step,
                                 "tower_steps": list(engine.tower_steps)}
        ck["lagrange"] = engine.lagrange.state_dict()
    if extra:
        ck.update(extra)
    torch.save(ck, path)


def load_checkpoint(path_or_dict, model, engine=None, drop_critic_tsfm: bool = False):
    ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, str) else path_or_dict
    sd = ck["model_state_dict"]
    if drop_critic_tsfm:   # RL -> RL init of a fresh reward critic (allenact_dino_transformer.py:186-188)
        sd = {k: v for k, v in sd.items() if "critic_tsfm" not in k}
    res = model.load_state_dict(sd, strict=not drop_critic_tsfm)
    if engine is not None and "optimizer_state" in ck:
        ar = model.arena
        ar.flat_m.copy_(ck["optimizer_state"]["exp_avg"]); ar.flat_v.copy_(ck["optimizer_state"]["exp_avg_sq"])
        engine.opt_step = int(ck["optimizer_state"]["step"])
        engine.tower_steps = [int(x) for x in ck["optimizer_state"].get("tower_steps", [engine.opt_step] * 3)]
        if "lagrange" in ck:
            engine.lagrange.load_state_dict(ck["lagrange"])
    return res
