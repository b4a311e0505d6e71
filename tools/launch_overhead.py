#!/usr/bin/env python3
"""CPU cost of one kernel launch through the Python binding (tiny problems: the GPU is never the limit)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd import ops
from safevla_amd._lib import lib
dev = "cuda"
x = torch.randn(64, 512, device=dev).to(torch.bfloat16); g = torch.ones(512, device=dev); b = torch.zeros(512, device=dev)
W = torch.randn(512, 512, device=dev).to(torch.bfloat16); out = torch.empty(64, 512, device=dev, dtype=torch.bfloat16)
def bench(name, f, n=3000):
    for _ in range(50): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:46s}: issue {1e6*(t1-t0)/n:6.2f} us/call   (drain {1e6*(t2-t1)/n:5.2f})")
bench("torch.empty(64,512,bf16)", lambda: torch.empty(64, 512, device=dev, dtype=torch.bfloat16))
bench("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
bench("x.data_ptr() x8", lambda: [x.data_ptr() for _ in range(8)])
bench("ops.norm_fwd (allocs y, no stats)", lambda: ops.norm_fwd(x, g, b, 1e-5, 64, save_stats=False))
bench("ops.gemm_nt (allocs out)", lambda: ops.gemm_nt(x, W, 64, 512, 512))
bench("ops.gemm_nt (out=)", lambda: ops.gemm_nt(x, W, 64, 512, 512, out=out))
st = torch.cuda.current_stream().cuda_stream
fn = lib().cdll.svla_gemm_nt_bf16
args = (x.data_ptr(), 512, W.data_ptr(), 512, None, None, 0, None, 0, out.data_ptr(), 512, 64, 512, 512, 0, 0, 1.0, None, None, None, st)
bench("raw ctypes call svla_gemm_nt_bf16", lambda: fn(*args))
