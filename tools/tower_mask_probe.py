#!/usr/bin/env python3
"""Acting step (64 envs, recorded plans) with the three towers' streams restricted to disjoint CU sets (hipExtStreamCreateWithCUMask) against
the unmasked three streams: does a tower stop waiting behind the other towers' chip-filling kernels?   python tools/tower_mask_probe.py [steps]"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = "/tmp/whereami.so"
if not os.path.exists(SO):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tools", "probes", "whereami_probe.hip"), "-o", SO], check=True)
lib = ctypes.CDLL(SO)
NCU = torch.cuda.get_device_properties(0).multi_processor_count
NW = (NCU + 31) // 32


def masked_stream(bits):
    words = [0] * NW
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (ctypes.c_uint32 * NW)(*words)
    s = ctypes.c_void_p()
    rc = lib.make_masked_stream(ctypes.byref(s), NW, arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
    return torch.cuda.ExternalStream(s.value)


dev = torch.device("cuda")
B, n = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 60
MASKS = {
    "unmasked": None,
    "bit % 3": [[b for b in range(NCU) if b % 3 == k] for k in range(3)],
    "(bit // 8) % 3": [[b for b in range(NCU) if (b // 8) % 3 == k] for k in range(3)],
    "contiguous thirds": [[b for b in range(NCU) if b * 3 // NCU == k] for k in range(3)],
    "actor half, critics quarter": [[b for b in range(NCU) if b % 4 < 2], [b for b in range(NCU) if b % 4 == 2], [b for b in range(NCU) if b % 4 == 3]],
    "overlapping two-thirds": [[b for b in range(NCU) if b % 3 != k] for k in range(3)],
}
for name, sets in MASKS.items():
    torch.manual_seed(1234)
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
    st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=n + 8, B=B, L=12, task="PickUp", seed=1234), device=dev)
    step_in = lambda t: ({k: v[t:t + 1] for k, v in st.observations.items()}, None, st.prev_actions[t:t + 1], st.masks[t:t + 1])
    for t in m.towers:
        t.time_step_counter, t._kv = 0, None
    if sets is not None:
        m._tower_streams = [masked_stream(s) for s in sets]
    m.enable_acting_plans(True)
    with torch.no_grad():
        for t in range(4):
            out = m(*step_in(t))
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            for tw in m.towers:
                pass
            t0 = time.perf_counter()
            for t in range(4, 4 + n // 3):
                out = m(*step_in(t + rep * (n // 3)))
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / (n // 3))
    print(f"{name:32s} {1e3 * best:.3f} ms per step, {B / best:.0f} env-steps/s  logits sum {float(out[0].distributions.logits.float().sum()):.4f}", flush=True)
    del m, st
