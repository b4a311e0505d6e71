#!/usr/bin/env python3
"""fp8 attention kernels vs exact fp32 attention and vs the bf16 kernels on the same inputs: relative Frobenius errors and cosines of O, dQ, dK, dV
(the numbers quoted in DESIGN.md section 5 and in tests/test_fp8_attention_gpu.py)."""
import sys; sys.path.insert(0, "/root/repo")
import torch
from safevla_amd import ops
sys.path.insert(0, "/root/repo/tests")
import test_fp8_attention_gpu as T
for S in (64, 181, 233):
    rows, H, scale = 4, 8, 0.125
    qkv, do = T._case(rows, S, H, 100 + S)
    o, dq, dk, dv = T._exact(qkv, do, rows, S, H, scale)
    d = qkv.cuda().bfloat16()
    f8 = ops.attn_fp8_quant(d, 3 * H * 64, rows, S, H)
    out, lse = ops.attn_fp8_fwd(f8, scale)
    dqkv = torch.zeros_like(d)
    ops.attn_fp8_bwd(f8, out, lse, do.cuda().bfloat16(), dqkv, dqkv[:, 512:], dqkv[:, 1024:], 1536, scale)
    # bf16 kernels for comparison
    ob, lb = ops.attn_fwd(d, d[:, 512:], d[:, 1024:], 1536, rows, S, H, scale)
    db = torch.zeros_like(d)
    ops.attn_bwd(d, d[:, 512:], d[:, 1024:], 1536, ob, 512, lb, do.cuda().bfloat16(), 512, db, db[:, 512:], db[:, 1024:], 1536, rows, S, H, scale)
    torch.cuda.synchronize()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    cos = lambda a, b: torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
    print(f"S={S}: O rel fp8 {rel(T._heads(out.float().cpu(), rows, S, H), o.detach()):.4f} bf16 {rel(T._heads(ob.float().cpu(), rows, S, H), o.detach()):.4f}")
    for i, (n, w) in enumerate((("dQ", dq), ("dK", dk), ("dV", dv))):
        g8 = T._heads(dqkv[:, i*512:(i+1)*512].float().cpu(), rows, S, H); gb = T._heads(db[:, i*512:(i+1)*512].float().cpu(), rows, S, H)
        print(f"   {n}: fp8 rel {rel(g8, w):.4f} cos {cos(g8, w):.5f} | bf16 rel {rel(gb, w):.4f} cos {cos(gb, w):.5f}")
