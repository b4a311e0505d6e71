#!/usr/bin/env python3
"""A/B of the attention backward: single-pass kernel vs the dQ + dK/dV kernel pair (fusion-encoder shape), with a result comparison."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops

def t_ms(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (R, S, drop) in [(int(os.environ.get("AB_ROWS", 8192)), 181, None), (4096, 233, None), (8192, 181, ops.Dropout(77, 3, 0.1)), (2048, 100, None), (4096, 50, None)]:
    qkv = (torch.randn(R * S, 1536, device="cuda") * 0.5).to(torch.bfloat16)
    kw = dict(drop=drop) if drop is not None else {}
    out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, **kw)
    do = torch.randn_like(out)
    res = {}
    for two in (1, 0):
        ops.attn_bwd_two_pass(two)
        dqkv = torch.zeros_like(qkv)
        f = lambda: ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125, **kw)
        ms = t_ms(f)
        res[two] = (ms, dqkv.float())
    ops.attn_bwd_two_pass(0)
    a, b = res[1][1], res[0][1]
    err = (a - b).abs().max().item(); ref = a.abs().max().item()
    cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    print(f"R={R} S={S} drop={'yes' if drop else 'no'}: two-pass {res[1][0]:.3f} ms, single-pass {res[0][0]:.3f} ms; max |diff| {err:.3e} (max |grad| {ref:.3e}), cosine {cos:.7f}")

# fp8 variant (BASELINE config 5) next to the bf16 kernels, S = 233
for (R, S) in [(4096, 233), (8192, 181)]:
    qkv = (torch.randn(R * S, 1536, device="cuda") * 0.5).to(torch.bfloat16)
    out, lse = ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125)
    do = torch.randn_like(out); dqkv = torch.zeros_like(qkv)
    fb = t_ms(lambda: ops.attn_fwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, R, S, 8, 0.125, out=out))
    bb = t_ms(lambda: ops.attn_bwd(qkv, qkv[:, 512:], qkv[:, 1024:], 1536, out, 512, lse, do, 512, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, R, S, 8, 0.125))
    f8 = ops.attn_fp8_quant(qkv, 1536, R, S, 8)
    tq = t_ms(lambda: ops.attn_fp8_quant(qkv, 1536, R, S, 8))
    o8, l8 = ops.attn_fp8_fwd(f8, 0.125)
    f8f = t_ms(lambda: ops.attn_fp8_fwd(f8, 0.125, out=o8))
    f8b = t_ms(lambda: ops.attn_fp8_bwd(f8, o8, l8, do, dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, 0.125))
    print(f"R={R} S={S}: bf16 fwd {fb:.3f} ms bwd {bb:.3f} ms | fp8 quant {tq:.3f} ms fwd {f8f:.3f} ms bwd (prep + kernel) {f8b:.3f} ms")
