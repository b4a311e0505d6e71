"""TEST INFRASTRUCTURE ONLY (oracle/): fp32 torch restatement of the fp8 attention kernels of safevla_amd/csrc/attn_fp8.hip with the casts at the
kernels' exact quantisation points -- forward AND backward (the backward had no quantisation-aware reference).

Reference op: nn.MultiheadAttention of the fusion encoder (architecture/models/allenact_transformer_models/allenact_dino_transformer.py:545-552,
702-708), computed the way BASELINE.json configs[4] ("fp8 MFMA attention") names it.  Parity status: the ARITHMETIC is pinned by construction (exact
fp32 attention is the first rung of the ladder in tests/test_fp8_attention_gpu.py); this file pins the QUANTISATION: per [S, 64] head slice
  Q, K, V  -> e4m3 of x * (448 / amax)            (dequantisation multipliers sq, sk, sv = amax / 448)
  dO       -> e5m2 of x * (16384 / amax)          (sg = amax / 16384)
  P        -> e4m3 of keep * P * 256 * drop_scale (P = exp(sq sk scale Q8.K8^T - lse) in fp32)
  dS       -> e5m2 of clamp(P * tt * 2^-13, +-49152),  tt = keep * (G8.V8^T) * drop_scale - D / (sg sv),  D = rowsum(dO * O)
  dV = (P8^T G8) sg / 256,   dK = (dS8^T Q8) sq sg sv scale 2^13,   dQ = (dS8 K8) sk sg sv scale 2^13
with every product accumulated in fp32.  Tensors are [rows, H, S, 64]."""
import torch

E4M3, E5M2 = torch.float8_e4m3fn, torch.float8_e5m2


def quant_slice(t, dtype, target):
    """(values on the fp8 grid as fp32, dequantisation multiplier [rows, H, 1, 1]); the kernel multiplies by target / amax"""
    amax = t.abs().amax(dim=(-1, -2), keepdim=True)
    inv = torch.where(amax > 0, target / amax, torch.zeros_like(amax))
    return (t * inv).to(dtype).float(), torch.where(amax > 0, amax / target, torch.zeros_like(amax))


def fwd(q, k, v, scale, keep=None, drop_scale=1.0):
    (q8, sq), (k8, sk), (v8, sv) = [quant_slice(t, E4M3, 448.0) for t in (q, k, v)]
    x = (q8 @ k8.transpose(-1, -2)) * (sq * sk * scale)
    mx = x.amax(-1, keepdim=True)
    pu = torch.exp(x - mx)
    pk = pu if keep is None else torch.where(keep, pu * drop_scale, torch.zeros_like(pu))
    p8 = (pk * 256.0).to(E4M3).float()
    o = (p8 @ v8) * sv / (256.0 * pu.sum(-1, keepdim=True))
    return o, (mx + torch.log(pu.sum(-1, keepdim=True))).squeeze(-1)


def bwd(q, k, v, o, lse, do, scale, keep=None, drop_scale=1.0):
    """o, lse: the forward's outputs as the backward kernel receives them (bf16 O, fp32 LSE); do: bf16-exact upstream gradient"""
    (q8, sq), (k8, sk), (v8, sv) = [quant_slice(t, E4M3, 448.0) for t in (q, k, v)]
    g8, sg = quant_slice(do, E5M2, 16384.0)
    D = (do * o).sum(-1, keepdim=True)
    gv = sg * sv
    p = torch.exp((q8 @ k8.transpose(-1, -2)) * (sq * sk * scale) - lse.unsqueeze(-1))
    dp = g8 @ v8.transpose(-1, -2)
    nd = -D / gv
    if keep is None:
        tt, pv = dp + nd, p * 256.0
    else:
        tt = torch.where(keep, dp * drop_scale, torch.zeros_like(dp)) + nd
        pv = torch.where(keep, p * (256.0 * drop_scale), torch.zeros_like(p))
    p8 = pv.to(E4M3).float()
    d8 = (p * tt * 2.0 ** -13).clamp(-49152.0, 49152.0).to(E5M2).float()
    dv = (p8.transpose(-1, -2) @ g8) * (sg / 256.0)
    dk = (d8.transpose(-1, -2) @ q8) * (sq * gv * scale * 8192.0)
    dq = (d8 @ k8) * (sk * gv * scale * 8192.0)
    return dq, dk, dv
