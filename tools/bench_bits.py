#!/usr/bin/env python3
"""ReLU bit-mask vs bf16 activation mask in the FFN GEMMs (fusion shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
M = 8192 * 181
def t_ms(f, n=10, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(M, 512, device="cuda").to(torch.bfloat16); W1 = (torch.randn(2048, 512, device="cuda") * 0.05).to(torch.bfloat16)
b1 = torch.randn(2048, device="cuda") * 0.1
h = torch.empty(M, 2048, device="cuda", dtype=torch.bfloat16); bits = torch.empty(ops.relu_bits_bytes(M, 2048), device="cuda", dtype=torch.uint8)
print("ffn1 fwd relu           : %.3f ms" % t_ms(lambda: ops.gemm_nt(x, W1, M, 2048, 512, bias=b1, act=ops.ACT_RELU, out=h)))
print("ffn1 fwd relu + bits_out: %.3f ms" % t_ms(lambda: ops.gemm_nt(x, W1, M, 2048, 512, bias=b1, act=ops.ACT_RELU, out=h, relu_bits_out=bits)))
dy = torch.randn(M, 512, device="cuda").to(torch.bfloat16); W2t = (torch.randn(2048, 512, device="cuda") * 0.05).to(torch.bfloat16)
dh = torch.empty(M, 2048, device="cuda", dtype=torch.bfloat16)
print("dx_ffn2 plain           : %.3f ms" % t_ms(lambda: ops.gemm_nt(dy, W2t, M, 2048, 512, out=dh)))
print("dx_ffn2 relu_mask=h     : %.3f ms" % t_ms(lambda: ops.gemm_nt(dy, W2t, M, 2048, 512, out=dh, relu_mask=h)))
print("dx_ffn2 relu_bits       : %.3f ms" % t_ms(lambda: ops.gemm_nt(dy, W2t, M, 2048, 512, out=dh, relu_bits=bits)))
