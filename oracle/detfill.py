"""TEST INFRASTRUCTURE ONLY -- deterministic, torch-RNG-independent weight fill.

The golden fixtures cannot ship 169 M parameters, so both sides (the reference imported in the
build container, and the oracle / HIP model under test) regenerate identical weights from the
parameter *names*: each tensor is drawn from ``numpy.random.RandomState`` (a frozen legacy stream)
seeded by ``crc32(name) ^ seed``.  Scales are chosen so activations stay O(1) through the network
and every parameter visibly influences the outputs (a stricter parity probe than the reference's
own tiny-init defaults).
"""
import zlib

import numpy as np
import torch


def _draw(name: str, shape, seed: int):
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
    n = int(np.prod(shape)) if len(shape) else 1
    if len(shape) <= 1:
        if name.endswith("weight"):  # norm scales
            v = 1.0 + 0.1 * rs.standard_normal(n)
        elif name.endswith("bias"):
            v = 0.05 * rs.standard_normal(n)
        else:  # fusion / camera tokens
            v = 0.1 * rs.random_sample(n)
    else:
        is_table = ("embed" in name) or ("shared" in name) or ("relative_attention_bias" in name)
        if is_table:
            v = 0.5 * rs.standard_normal(n)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = rs.standard_normal(n) / np.sqrt(fan_in)
    return v.astype(np.float32).reshape(shape)


def tower_local_name(name: str) -> str:
    for p in ("critic_tsfm.", "c_critic_tsfm."):
        if name.startswith(p):
            return name[len(p):]
    return name


def fill_state_dict(module: torch.nn.Module, seed: int = 0, skip=("div_term",), share_t5=True) -> None:
    """Overwrite every parameter/buffer of ``module`` in place from its state_dict name.

    ``share_t5``: the frozen T5 encoder is loaded from the same ``t5-small`` checkpoint in all three
    towers of the reference (allenact_dino_transformer.py:506-508), so its weights are filled
    identically across towers; every other tensor is tower-specific.
    """
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if any(s in name for s in skip):
                continue
            if not t.dtype.is_floating_point:
                continue
            key = name
            if "text_encoder." in name:
                if share_t5:
                    key = tower_local_name(name)
                # tied embedding: both names must get the same values
                key = key.replace("encoder.embed_tokens.weight", "shared.weight")
            t.copy_(torch.from_numpy(_draw(key, tuple(t.shape), seed)))


def grad_probe(name: str, g: torch.Tensor):
    """(L2 norm, projection on a fixed pseudo-random direction) -- a 2-float checksum of a gradient."""
    rs = np.random.RandomState((zlib.crc32(("probe:" + name).encode())) & 0x7FFFFFFF)
    d = torch.from_numpy(rs.standard_normal(g.numel()).astype(np.float32))
    gf = g.detach().double().reshape(-1).cpu()
    return float(gf.norm()), float((gf * d.double()).sum() / np.sqrt(g.numel()))
