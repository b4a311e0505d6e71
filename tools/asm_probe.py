#!/usr/bin/env python3
"""run svla_probe (asmgen/probe_gen.py) and print what the hardware did"""
import ctypes, os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd._lib import lib
grid = 3
out = torch.zeros(4 * 65536 // 4, device="cuda", dtype=torch.int32)
src = torch.arange(16384, device="cuda", dtype=torch.int32)
srcb = torch.randn(256 * 8, device="cuda").to(torch.bfloat16)
src[: 256 * 4] = srcb.view(torch.int32)
ka = struct.pack("<QQQQ", out.data_ptr(), src.data_ptr(), int(os.environ.get("PROBE_STOP", "0")), 0)
buf = ctypes.create_string_buffer(ka, len(ka))
L = lib()
L.cdll.svla_asm_launch_raw.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
rc = L.cdll.svla_asm_launch_raw(b"svla_probe", buf, 32, grid, 256, None)
torch.cuda.synchronize()
print("rc", rc)
o = out.cpu().view(4, -1, 4)
print("record -1 (s2, v0, s3, wave) by lane:", [o[3][-64 + i].tolist() for i in (0, 1, 2, 63)])
if os.environ.get("PROBE_STOP"): sys.exit(0)
r0 = o[0][: grid * 256]
print("record 0 (s2, v0, wave, marker) for threads 0, 1, 64, 255, 256, 600:", [r0[i].tolist() for i in (0, 1, 64, 255, 256, 600)])
s_ = src.cpu().view(-1, 4)
print("record 1 (buffer_load -> AGPR) ok:", bool((o[1][:256] == s_[:256]).all()))
exp = torch.stack([s_[(t // 64) * 64 + ((t % 64) ^ 5)] for t in range(256)])
print("record 2 (global_load_lds saddr, M0 > 64 KiB) ok:", bool((o[2][:256] == exp).all()), o[2][:2].tolist(), exp[:2].tolist())
# MFMA: D[i][j] = sum_k A[i][k] B[k][j] with A-frag = B-frag data: lane l holds X[l&31][8(l>>5)+e]; D = X X^T; D reg r lane l: row (r&3)+8(r>>2)+4(l>>5), col l&31
X = torch.zeros(4, 32, 16)
sb = srcb.cpu().float().view(256, 8)
for w in range(4):
    for l in range(64):
        X[w, l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = sb[w * 64 + l]
D = X @ X.transpose(1, 2)
got = o[3][:256].view(torch.float32).view(4, 64, 4)
ok = True
for w in range(4):
    for l in range(64):
        for r in range(4):
            ok &= abs(got[w, l, r].item() - D[w, (r & 3) + 4 * (l >> 5), l & 31].item()) < 1e-2
print("record 3 (MFMA, B operand in AGPR) ok:", ok)
