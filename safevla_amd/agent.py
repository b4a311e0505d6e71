"""Evaluation / rollout-time agent on the MI355X acting path.

Mirrors ``InferenceAgentVIDA`` (/root/reference/architecture/models/allenact_transformer_models/inference_agent.py:74-296) behind
``AbstractAgent`` (/root/reference/architecture/agent.py:5-51): ``reset()``, ``get_action_list()`` and
``get_action(frame, goal_spec) -> (action_str, action_probs)``.  One call = uint8 frames -> frozen DINOv2 ViT preprocessor
(``preproc.DinoViTPreprocessor``) -> single-step 3-tower forward with per-tower llama KV caches (``model`` acting path) ->
sample / mode.  The rollout storage is used exactly as the reference uses it (``initialize`` on the first step of a task, ``add``
with dummy value/reward fields afterwards, ``agent_input_for_next_step``), so the prev-action / mask / time-step plumbing is the
update path's own.
"""
import json
import os
import warnings
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import checkpoint
from .preproc import DinoViTPreprocessor
from .storage import RolloutStorage
from .text import str_to_bytes

# Stretch action vocabulary in the order of the policy head (utils/constants/stretch_initialization_utils.py:145-166,
# short names utils/type_utils.py:55-74; the ACTION_DICT / LONG_ACTION_NAME environment overrides are honoured like upstream)
ALL_STRETCH_ACTIONS = ["m", "r", "l", "b", "end", "sub_done", "ls", "rs", "p", "zm", "zp", "yp", "ym", "wp", "wm", "yms", "zms",
                       "zps", "yps", "d"]
STRETCH_LONG_NAMES = {"m": "move_ahead", "r": "rotate_right", "l": "rotate_left", "b": "move_back", "end": "done",
                      "sub_done": "sub_done", "ls": "rotate_left_small", "rs": "rotate_right_small", "p": "pickup",
                      "zm": "move_arm_in", "zp": "move_arm_out", "yp": "move_arm_up", "ym": "move_arm_down", "wp": "wrist_open",
                      "wm": "wrist_close", "yms": "move_arm_down_small", "zms": "move_arm_in_small", "zps": "move_arm_out_small",
                      "yps": "move_arm_up_small", "d": "dropoff"}


class AbstractAgent:
    def reset(self) -> None:
        raise NotImplementedError

    def get_action_list(self) -> List[str]:
        raise NotImplementedError

    def get_action(self, observations: Dict[str, Any], goal: str) -> Tuple[str, Any]:
        raise NotImplementedError


class InferenceAgentVIDA(AbstractAgent):
    num_evaluated_traj = 0

    def __init__(self, actor_critic, device="cuda", greedy_sampling: bool = False, nav_preprocessor=None, manip_preprocessor=None,
                 steps_before_rollout_refresh: int = 64, generator: Optional[torch.Generator] = None):
        self.actor_critic = actor_critic
        self.device = torch.device(device)
        self.greedy_sampling = greedy_sampling
        u = actor_critic.uuids
        self.nav_pre = nav_preprocessor or DinoViTPreprocessor("rgb_raw", u["nav"], device=device)
        self.manip_pre = manip_preprocessor or DinoViTPreprocessor("manipulation_rgb_raw", u["manip"], device=device)
        if manip_preprocessor is None and nav_preprocessor is None:
            self.manip_pre.vit = self.nav_pre.vit            # one frozen ViT serves both cameras
        self.steps_before_rollout_refresh = steps_before_rollout_refresh
        self.rollout_storage = RolloutStorage(steps_before_rollout_refresh, device=device, store_tokens=True,
                                              nav_uuid=u["nav"], manip_uuid=u["manip"])
        self.generator = generator
        self.has_initialized = False
        self.memory = None
        self.steps_taken_in_task = 0
        self.last_action_flat = None

    IL_VIT_PREFIX = "model.visual_encoder.image_encoder.model."      # where a Lightning IL checkpoint keeps its DINOv2 weights

    @classmethod
    def build_agent(cls, actor_critic, device="cuda", greedy_sampling: bool = False, ckpt_path: Optional[str] = None,
                    spiece_model: Optional[str] = None, vit_weights: Optional[str] = None, **kw):
        """ckpt formats auto-detected like upstream (:128-165): Lightning ``state_dict`` (IL), AllenAct ``model_state_dict``, or a
        bare state dict.

        The reference pulls two more assets from the network that a checkpoint does not (or only partly) contain: the ``t5-small``
        sentencepiece vocabulary and the frozen DINOv2 ViT-S/14 weights (torch.hub).  ``spiece_model`` = path of ``spiece.model``;
        ``vit_weights`` = path of a DINOv2 ViT-S/14 ``state_dict`` (hub names).  A Lightning IL checkpoint that carries its image
        encoder (``model.visual_encoder.image_encoder.model.*``) fills the ViT from there.  Running real weights on the offline
        stand-ins (word-hash token ids, random-init ViT) produces meaningless actions, so that combination WARNS loudly."""
        vit_sd = None
        if ckpt_path is not None:
            ckpt = torch.load(ckpt_path, map_location="cpu")
            if "state_dict" in ckpt:
                checkpoint.load_pl_ckpt_allenact(actor_critic, ckpt)
                actor_critic.sync_weights()
                vit_sd = {k[len(cls.IL_VIT_PREFIX):]: v for k, v in ckpt["state_dict"].items() if k.startswith(cls.IL_VIT_PREFIX)} or None
            elif "model_state_dict" in ckpt:
                actor_critic.load_state_dict(ckpt["model_state_dict"], strict=False)
            elif any(k.startswith(("visual_encoder.", "actor.", "decoder.")) for k in ckpt):
                actor_critic.load_state_dict(ckpt, strict=False)
            else:
                raise ValueError(f"Unknown checkpoint format; found keys {list(ckpt.keys())[:10]}")
        if spiece_model is not None:
            from .text import GoalTokenizer

            actor_critic.tokenizer = GoalTokenizer(spiece_model)
            actor_critic._goal_cache.clear()
        if vit_weights is not None:
            vit_sd = torch.load(vit_weights, map_location="cpu")
        actor_critic.eval()     # AllenAct's InferenceAgent runs the policy in eval mode (mode="test", inference_agent.py:98-102)
        agent = cls(actor_critic, device=device, greedy_sampling=greedy_sampling, **kw)
        if vit_sd is not None:
            missing, unexpected = agent.nav_pre.vit.load_state_dict(vit_sd, strict=False)
            if missing:
                warnings.warn(f"DINOv2 weights: {len(missing)} tensors not found (e.g. {missing[:3]}); those stay random-init")
            agent.nav_pre.vit.sync()
            if agent.manip_pre.vit is not agent.nav_pre.vit:
                agent.manip_pre.vit.load_state_dict(vit_sd, strict=False)
                agent.manip_pre.vit.sync()
        if ckpt_path is not None:
            stand_ins = []
            if getattr(actor_critic.tokenizer, "_sp", None) is None:
                stand_ins.append("word-hash goal tokenizer (pass spiece_model=<t5-small spiece.model>)")
            if vit_sd is None and kw.get("nav_preprocessor") is None:
                stand_ins.append("random-init DINOv2 ViT (pass vit_weights=<dinov2_vits14 state_dict>)")
            if stand_ins:
                warnings.warn("checkpoint loaded but offline stand-ins are active: " + "; ".join(stand_ins) +
                              " -- the policy's inputs do not match what it was trained on, actions are meaningless", RuntimeWarning)
        agent.reset()
        return agent

    # ---- AbstractAgent ------------------------------------------------------------------------------------------------------
    def reset(self):
        if self.has_initialized:
            self.rollout_storage.after_updates()
        self.steps_taken_in_task = 0
        type(self).num_evaluated_traj += 1
        self.traj_index = type(self).num_evaluated_traj
        self.memory = None

    def get_action_list(self) -> List[str]:
        if os.getenv("ACTION_DICT") is not None:
            return list(json.load(open(os.getenv("ACTION_DICT"), "r")).keys())
        if os.getenv("LONG_ACTION_NAME") is not None and bool(int(os.getenv("LONG_ACTION_NAME"))):
            return [STRETCH_LONG_NAMES[a] for a in ALL_STRETCH_ACTIONS]
        return list(ALL_STRETCH_ACTIONS)

    def get_action(self, frame: Dict[str, Any], goal_spec: str) -> Tuple[str, torch.Tensor]:
        """frame: the evaluator's sensor dict (``raw_navigation_camera`` [, ``raw_manipulation_camera``, ``an_object_is_in_hand``])."""
        optional = {"raw_manipulation_camera": "manipulation_rgb_raw", "an_object_is_in_hand": "an_object_is_in_hand"}
        sensors = {dst: frame[src] for src, dst in optional.items() if src in frame}
        sensors.update(rgb_raw=frame["raw_navigation_camera"], natural_language_spec=str_to_bytes(goal_spec, 1000),
                       time_step=self.steps_taken_in_task, traj_index=self.traj_index)
        return self.act(sensors, goal_spec)

    # ---- one acting step ----------------------------------------------------------------------------------------------------
    def _batch(self, observations: Dict[str, Any]) -> Dict[str, torch.Tensor]:
        dev, u = self.device, self.actor_critic.uuids
        as_u8 = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev).reshape((1,) + tuple(np.shape(a)))
        nav = as_u8(observations["rgb_raw"])
        manip = as_u8(observations["manipulation_rgb_raw"]) if "manipulation_rgb_raw" in observations else torch.zeros_like(nav)
        return {
            u["nav"]: self.nav_pre.process({"rgb_raw": nav}),
            u["manip"]: self.manip_pre.process({"manipulation_rgb_raw": manip}),
            u["goal"]: torch.as_tensor(np.asarray(observations["natural_language_spec"], dtype=np.uint8)).to(dev).reshape(1, -1),
            u["time"]: torch.tensor([int(observations["time_step"])], device=dev, dtype=torch.int64),
            u["traj"]: torch.tensor([int(observations["traj_index"])], device=dev, dtype=torch.int64),
            u["hand"]: torch.tensor([int(np.asarray(observations.get("an_object_is_in_hand", 0)).reshape(-1)[0])], device=dev, dtype=torch.int64),
        }

    def _remember(self, obs_batch: Dict[str, torch.Tensor]):
        """Feed the step's observations to the rollout storage the way the update path's storage is fed: a task's first step
        (re)initialises it, later steps append with the sampled previous action; value / reward / cost fields are unused placeholders."""
        st = self.rollout_storage
        if self.steps_taken_in_task > 0:
            zero = torch.zeros((1, 1), device=self.device)
            st.add(observations=obs_batch, memory=self.memory, actions=self.last_action_flat, action_log_probs=zero, value_preds=zero,
                   rewards=zero, costs=zero, c_value_preds=zero, masks=torch.ones((1, 1), device=self.device))   # one task until reset(): never "done"
            return
        self.has_initialized = True
        st.initialize(observations=obs_batch, num_samplers=1, recurrent_memory_specification=self.actor_critic.recurrent_memory_specification,
                      action_space=None)
        st.after_updates()

    @torch.no_grad()
    def act(self, observations: Dict[str, Any], goal_spec: str = "") -> Tuple[str, torch.Tensor]:
        self._remember(self._batch(observations))
        st = self.rollout_storage
        out, self.memory = self.actor_critic(**st.agent_input_for_next_step())
        dist = out.distributions
        sampled, greedy = dist.sample(generator=self.generator), dist.mode()
        self.last_action_flat = sampled.reshape(1)          # what the policy sees as "previous action" next step is the sampled action, also when acting greedily
        self.steps_taken_in_task += 1
        if st.step == st.T:                                 # storage full: roll the last slot to the front
            st.after_updates()
        idx = int((greedy if self.greedy_sampling else sampled).reshape(-1)[0])
        return self.get_action_list()[idx], dist.probs[0][0]
