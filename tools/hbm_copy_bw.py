import torch, time
x = torch.empty(1482752, 2048, device="cuda", dtype=torch.bfloat16).normal_()
y = torch.empty_like(x)
for n in (512, 2048):
    a, b = x[:, :n].contiguous(), torch.empty(1482752, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"copy [{a.shape[0]}, {n}] bf16: {ms:.3f} ms  {2*a.numel()*2/ms/1e6:.0f} GB/s (read+write)")
    c = torch.empty_like(a)
    e0.record()
    for _ in range(10): torch.add(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"add  [{a.shape[0]}, {n}] bf16: {ms:.3f} ms  {3*a.numel()*2/ms/1e6:.0f} GB/s (2 reads + 1 write)")
# pure write / pure read
z = torch.empty(1482752, 2048, device="cuda", dtype=torch.bfloat16)
for _ in range(3): z.fill_(1.0)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): z.fill_(1.0)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"fill [{z.shape[0]}, 2048] bf16: {ms:.3f} ms  {z.numel()*2/ms/1e6:.0f} GB/s (write only)")
e0.record()
for _ in range(10): s_ = z.sum(dtype=torch.float32)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"sum  [{z.shape[0]}, 2048] bf16: {ms:.3f} ms  {z.numel()*2/ms/1e6:.0f} GB/s (read only)")
