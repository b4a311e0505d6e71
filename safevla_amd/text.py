"""Goal-text host path: byte strings -> T5 token ids.

Mirrors the host half of ``DinoTxGoalEncoder.distribute_target``
(/root/reference/architecture/models/allenact_transformer_models/allenact_dino_transformer.py:591-603)
and ``convert_byte_to_string`` (/root/reference/utils/string_utils.py:15-18).

The reference tokenises with HuggingFace ``AutoTokenizer.from_pretrained("t5-small")`` (a sentencepiece
model downloaded from the hub).  No network here, so:
  * if a sentencepiece ``spiece.model`` path is supplied, it is used (ids identical to t5-small);
  * otherwise a deterministic word-hash vocabulary stands in (synthetic ids in [3, 32000), EOS=1, PAD=0) --
    same shapes/ranges, used for synthetic benchmarks and parity fixtures on both sides.
"""
import zlib
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

PAD_ID, EOS_ID = 0, 1
VOCAB = 32128


def bytes_to_str(row: np.ndarray) -> str:
    return bytes(np.asarray(row, dtype=np.uint8)).split(b"\x00", 1)[0].decode()


def str_to_bytes(s: str, max_len: int = 1000) -> np.ndarray:
    """utils/string_utils.py:11-12 -- zero-padded fixed-width byte row."""
    return np.array([s], dtype=f"S{max_len}").view("uint8")


class _Encoding(dict):
    def to(self, device):
        return _Encoding({k: v.to(device) for k, v in self.items()})


class GoalTokenizer:
    def __init__(self, spiece_model: Optional[str] = None):
        self._sp = None
        if spiece_model is not None:
            import sentencepiece as spm

            self._sp = spm.SentencePieceProcessor(model_file=spiece_model)

    def encode(self, text: str) -> List[int]:
        if self._sp is not None:
            return list(self._sp.encode(text)) + [EOS_ID]
        return [3 + (zlib.crc32(w.lower().encode()) % (32000 - 3)) for w in text.split()] + [EOS_ID]

    def __call__(self, goals: Sequence[str], return_tensors: str = "pt", padding: bool = True) -> Dict[str, torch.Tensor]:
        enc = [self.encode(g) for g in goals]
        L = max(len(e) for e in enc)
        ids = torch.full((len(enc), L), PAD_ID, dtype=torch.int64)
        am = torch.zeros((len(enc), L), dtype=torch.int64)
        for i, e in enumerate(enc):
            ids[i, : len(e)] = torch.tensor(e, dtype=torch.int64)
            am[i, : len(e)] = 1
        return _Encoding(input_ids=ids, attention_mask=am)
