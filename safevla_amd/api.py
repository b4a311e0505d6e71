"""Output containers of the reference's model API (AllenAct [3P] types the reference returns from
``forward``: /root/reference/architecture/models/allenact_transformer_models/separate_actor_critic.py:27-37,
allenact_dino_transformer.py:470-475).  Same attribute names so losses / engines written against the
reference read these objects unchanged."""
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import torch


class CategoricalDistr:
    """AllenAct ``CategoricalDistr``: ``torch.distributions.Categorical(logits=...)`` semantics."""

    def __init__(self, logits: torch.Tensor):
        self.raw_logits = logits
        self.logits = logits - logits.logsumexp(dim=-1, keepdim=True)

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
