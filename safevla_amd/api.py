"""Output containers of the reference's model API (AllenAct [3P] types the reference returns from
``forward``: /root/reference/architecture/models/allenact_transformer_models/separate_actor_critic.py:27-37,
allenact_dino_transformer.py:470-475).  Same attribute names so losses / engines written against the
reference read these objects unchanged."""
from dataclasses import dataclass, field
from typing import Any, Dict, NamedTuple, Optional

import torch


class CategoricalDistr:
    """AllenAct ``CategoricalDistr``: ``torch.distributions.Categorical(logits=...)`` semantics."""

    def __init__(self, logits: torch.Tensor):
        self.raw_logits = logits
        # = logits - logits.logsumexp(-1, keepdim=True) (torch.distributions.Categorical's normalisation) as ONE fused kernel instead of eight small ones:
        # an acting step builds this object once per env step
        self.logits = torch.log_softmax(logits, dim=-1)

    @property
    def probs(self):
        return self.logits.exp()

    def log_prob(self, actions: torch.Tensor) -> torch.Tensor:
        a = actions if actions.dim() == self.logits.dim() else actions.unsqueeze(-1)
        return self.logits.gather(-1, a).squeeze(-1)

    def entropy(self) -> torch.Tensor:
        return -(self.logits.exp() * self.logits).sum(-1)

    def mode(self) -> torch.Tensor:
        return self.logits.argmax(dim=-1)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        p = self.probs
        return torch.multinomial(p.reshape(-1, p.shape[-1]), 1, generator=generator).reshape(p.shape[:-1])


@dataclass
class ActorCriticOutput:
    distributions: CategoricalDistr
    values: torch.Tensor
    extras: Dict[str, Any] = field(default_factory=dict)


@dataclass
class SafeActorCriticOutput:
    distributions: CategoricalDistr
    values: torch.Tensor
    c_values: torch.Tensor
    extras: Dict[str, Any] = field(default_factory=dict)


class SafeRLStepResult(NamedTuple):
    """What ``Task.step(action)`` returns in the reference (AllenAct-fork type [3P], constructed at
    /root/reference/tasks/abstract_task.py:369-381): the RLStepResult fields plus the per-step safety ``cost``
    (= number of triggered safety predicates, abstract_task.py:321-333)."""
    observation: Optional[Any]
    reward: Optional[float]
    cost: Optional[float]
    done: Optional[bool]
    info: Optional[Dict[str, Any]]

    def clone(self, new_info: Dict[str, Any]):
        return SafeRLStepResult(observation=new_info.get("observation", self.observation), reward=new_info.get("reward", self.reward),
                                cost=new_info.get("cost", self.cost), done=new_info.get("done", self.done), info=new_info.get("info", self.info))

    def merge(self, other: "SafeRLStepResult"):
        pick = lambda a, b: b if b is not None else a
        return SafeRLStepResult(*[pick(a, b) for a, b in zip(self, other)])


class Box(NamedTuple):
    """Stand-in for ``gym.spaces.Box`` (gym is not a dependency here): the ``observation_space`` attribute the reference's
    preprocessors expose (architecture/allenact_preprocessors/dino_preprocessors.py:90-99)."""
    low: float
    high: float
    shape: tuple
    dtype: str = "float32"
