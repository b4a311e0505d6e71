#!/usr/bin/env python3
"""How fast can the launch-bound acting step be ISSUED?  Records every C-ABI call of one single-step 3-tower forward (64 envs), then
replays the recorded (function, args) list from a tight Python loop (no tensor allocation, no wrappers) and from pre-converted ctypes
arguments.  Replayed steps reuse the recorded buffers / cache slot (timing only).  Prints ms per step for eager issue and both replays."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from safevla_amd._lib import lib
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.synth_env import SynthVectorEnv
from safevla_amd.storage import RolloutStorage

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
m.enable_acting_plans(False)          # measure the raw eager path and a raw replay of its launches (the model's own recorded path is the default otherwise)
if len(sys.argv) > 2 and sys.argv[2] == "serial":
    m.concurrent_towers = False       # towers one after the other on one stream (profiles/r02_replay_probe_sequential_towers.txt)
env = SynthVectorEnv(B, L=12, task="PickUp", seed=0, device=dev)
st = RolloutStorage(64, device=dev)
st.initialize(env.reset(), num_samplers=B)
inp = st.agent_input_for_next_step()
with torch.no_grad():
    for _ in range(3):
        m(inp["observations"], None, inp["prev_actions"], inp["masks"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m(inp["observations"], None, inp["prev_actions"], inp["masks"])
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20 * 1e3
    L = lib()
    L.recorder = []
    keep = m(inp["observations"], None, inp["prev_actions"], inp["masks"])
    rec, L.recorder = L.recorder, None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        for fn, a in rec:
            fn(*a)
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / 50 * 1e3
    pre = []
    for fn, a in rec:
        ca = tuple(t(v) if not isinstance(v, ctypes._SimpleCData) and v is not None and not hasattr(v, "_obj") else v for t, v in zip(fn.argtypes, a))
        pre.append((fn, ca))
    t0 = time.perf_counter()
    for _ in range(50):
        for fn, a in pre:
            fn(*a)
    torch.cuda.synchronize()
    replay_pre = (time.perf_counter() - t0) / 50 * 1e3
    # GPU-side time of the recorded step alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for fn, a in rec:
        fn(*a)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        for fn, a in rec:
            fn(*a)
    e1.record()
    torch.cuda.synchronize()
print(f"envs {B}: C-ABI calls per step {len(rec)}; eager {eager:.2f} ms/step ({B / eager * 1e3:.0f} env-steps/s); replay {replay:.2f} ms; "
      f"replay with pre-converted ctypes args {replay_pre:.2f} ms; GPU-side (events, back-to-back replays) {e0.elapsed_time(e1) / 10:.2f} ms")
