"""Worker of tests/test_dp_gpu.py: one data-parallel rank (torchrun) computing its environment shard's gradients on the HIP kernels
and SUM-all-reducing it tower range by tower range, asynchronously, exactly like PPOLagEngine does before Adam."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(dev, T, B):
    from oracle.detfill import fill_state_dict
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    torch.manual_seed(0)                        # the synthetic rollout samples its actions from torch's RNG: same rollout in every process
    m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
    fill_state_dict(m, seed=7)
    m.sync_weights()
    m.eval()                                    # deterministic: the shard split must not change the (row-indexed) dropout noise
    st, nxt, ep = fill_synthetic_rollout(m, SynthSpec(T=T, B=B, L=6, task="ObjectNav", seed=3), device=dev)
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    return m, PPOLagEngine(m, PPOLagConfig()), st


def main():
    from safevla_amd import parallel

    out, T, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, local, world = parallel.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    m, eng, st = build(dev, T, B)               # every rank builds the same GLOBAL rollout, then works on its env shard only
    if rank == 1:                               # make rank 1 differ (arena AND frozen T5), then let rank 0's replica win
        m.arena.flat_p.mul_(1.5)
        m.visual_encoder.text_encoder.shared.weight.mul_(0.5)
    before = (m.arena.flat_p.double().sum().item(), m.visual_encoder.text_encoder.shared.weight.double().sum().item())
    parallel.broadcast_model_(m)
    after = torch.tensor([m.arena.flat_p.double().sum().item(), m.visual_encoder.text_encoder.shared.weight.double().sum().item()], dtype=torch.float64, device=dev)
    chk = after.clone()
    torch.distributed.all_reduce(chk, op=torch.distributed.ReduceOp.MAX)
    assert torch.equal(chk, after), "replicas differ after broadcast_model_"
    assert rank == 0 or before != tuple(after.tolist())
    s, n = parallel.shard_envs(B, world, rank)
    n_total = parallel.global_count(T * n, dev)
    assert n_total == T * B
    m.zero_grad()
    eng._sums.zero_()
    # the engine's own exchange: each tower's gradient range goes to an asynchronous all-reduce right after that tower's backward
    eng._accumulate(st.batch_slice(s, s + n), n_total, 0.25, last=True)
    assert len(eng._pending) == 3
    for w in eng._pending:
        w.wait()
    eng._pending = []
    parallel.allreduce_sum_(eng._sums)
    if "--expect-backend" in sys.argv:
        assert torch.distributed.get_backend() == sys.argv[sys.argv.index("--expect-backend") + 1], torch.distributed.get_backend()
    if rank == 0:
        torch.save({"flat_g": m.arena.flat_g.cpu(), "sums": eng._sums.cpu(), "backend": torch.distributed.get_backend(), "world": world, "pending": 3}, out)
    parallel.barrier()


if __name__ == "__main__":
    main()
