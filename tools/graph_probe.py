import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout
dev = torch.device("cuda"); B = 32
m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=40, B=B, L=12, task="Fetch", seed=1), device=dev)
for mode in (True, False):
    m.train(mode)
    for t in m.towers: t.time_step_counter, t._kv = 0, None
    m.enable_acting_graphs(True)
    step_in = lambda t: ({k: v[t:t + 1] for k, v in st.observations.items()}, None, st.prev_actions[t:t + 1], st.masks[t:t + 1])
    with torch.no_grad():
        for t in range(4): m(*step_in(t))
        g = list(m._acting_graphs.values())[0].graph
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"train={mode}: graph.replay alone {1e3*(t1-t0)/20:.3f} ms")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(4, 28): m(*step_in(t))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(f"train={mode}: full step (prepare + copies + replay) {1e3*(t1-t0)/24:.3f} ms")
    m.enable_acting_graphs(False)
