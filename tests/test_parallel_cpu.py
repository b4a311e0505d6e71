"""N > 1 path on CPU: two gloo ranks shard the environments, each computes gradients of the (oracle) policy on its
shard with the global 1/N normalisation, SUM-all-reduce the flat gradient and the cost accumulator through
safevla_amd.parallel -- the result must equal the single-process full-batch gradient and Jc."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_policy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 21))


def _loss_sum(net, x, batch, lam, n_total):
    from oracle import ref_loss

    out = net(x)
    logits, values = out[..., :20], out[..., 20:]
    total, _ = ref_loss.safe_ppo_log_grad(logits, values, batch, lam)
    return total * (x.shape[0] * x.shape[1]) / n_total        # mean over local rows -> sum / global rows


def _make(T=6, B=8):
    rs = np.random.RandomState(0)
    x = torch.from_numpy(rs.standard_normal((T, B, 16)).astype(np.float32))
    batch = {"actions": torch.from_numpy(rs.randint(0, 20, (T, B))), "old_action_log_probs": torch.full((T, B), -3.0),
             "adv_targ": torch.from_numpy(rs.standard_normal((T, B, 1)).astype(np.float32)),
             "c_adv_targ": torch.from_numpy(rs.standard_normal((T, B, 1)).astype(np.float32)),
             "returns": torch.from_numpy(rs.standard_normal((T, B, 1)).astype(np.float32)), "values": torch.zeros(T, B, 1)}
    return x, batch


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from safevla_amd import parallel

    r, _, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and parallel.is_dist()
    x, batch = _make()
    T, B = x.shape[:2]
    s, n = parallel.shard_envs(B, world, rank)
    net = _tiny_policy()
    n_total = parallel.global_count(T * n, "cpu")
    assert n_total == T * B
    loss = _loss_sum(net, x[:, s:s + n], {k: v[:, s:s + n] for k, v in batch.items()}, 0.37, n_total)
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    # the engine's exchange pattern: contiguous ranges of one flat buffer reduced asynchronously, waited for before the optimiser step
    cut = flat.numel() // 3
    handles = [parallel.allreduce_sum_async(flat[a:b]) for a, b in ((0, cut), (cut, 2 * cut), (2 * cut, flat.numel()))]
    for h in handles:
        h.wait()
    # the same exchange with bf16 on the wire (PPOLagConfig.grad_allreduce_dtype = "bf16", bench.py --grad-allreduce-bf16): the fp32 range comes back holding the
    # backend's sum of the ranks' bf16-rounded shards -- within bf16 rounding of the fp32 exchange, and identical on every rank
    local = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    wire = local.clone()
    hb = [parallel.allreduce_sum_async(wire[a:b], wire_dtype=torch.bfloat16) for a, b in ((0, cut), (cut, 2 * cut), (2 * cut, wire.numel()))]
    for h in hb:
        assert h.wait()
    assert wire.dtype == torch.float32 and torch.equal(wire, wire.to(torch.bfloat16).float())
    tol = 2.0 ** -7 * (local.abs() + (flat - local).abs()) + 1e-12          # two roundings to 8 significant bits + one of the sum
    assert bool(((wire - flat).abs() <= tol).all()), float((wire - flat).abs().max())
    every = [torch.zeros_like(wire) for _ in range(world)]
    dist.all_gather(every, wire)
    assert all(torch.equal(e, wire) for e in every)
    jc, n_ep = parallel.mean_episode_cost(3.0 * (rank + 1), 2.0, "cpu")
    # bench.py's collective pre-flight on a stand-in arena (three "tower ranges" of one flat buffer) + the per-rank clock gather
    class _Arena:
        flat_g = torch.zeros(3000)
        tower_ranges = [(0, 1024), (1024, 2048), (2048, 3000)]
    class _M:
        arena = _Arena()
    pre = parallel.preflight(_M(), "cpu")
    assert pre["ranks"] == world and pre["tower_ranges_checked"] == 3 and float(_Arena.flat_g.abs().sum()) == 0.0, pre
    assert parallel.gather_floats(10.0 + rank, "cpu") == [10.0 + i for i in range(world)]
    parallel.barrier()
    if rank == 0:
        q.put((flat.numpy(), jc, n_ep))
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    flat, jc, n_ep = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, batch = _make()
    net = _tiny_policy()
    _loss_sum(net, x, batch, 0.37, x.shape[0] * x.shape[1]).backward()
    want = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    np.testing.assert_allclose(flat, want, rtol=1e-5, atol=1e-7)
    assert n_ep == 4.0 and abs(jc - (3.0 + 6.0) / 4.0) < 1e-12


def test_forced_single_rank_goes_through_the_backend():
    """SVLA_FORCE_DIST=1 (the 1-GPU boxes' way to execute the RCCL path, tests/test_dp_gpu.py): a single rank still initialises a process group and
    every helper of safevla_amd.parallel calls the backend's collectives; here on gloo."""
    import subprocess

    code = ("import torch\nfrom safevla_amd import parallel\n"
            "r, l, w = parallel.init_from_env(backend='gloo')\n"
            "assert (r, w) == (0, 1) and parallel.is_dist() and torch.distributed.get_backend() == 'gloo'\n"
            "t = torch.arange(6.0)\nh = parallel.allreduce_sum_async(t)\nassert h.wait() and t.tolist() == [0, 1, 2, 3, 4, 5]\n"
            "assert parallel.global_counts([3, 9], 'cpu') == [3, 9] and parallel.mean_episode_cost(4.0, 2.0, 'cpu') == (2.0, 2.0)\n"
            "class A:\n    flat_g = torch.zeros(30)\n    tower_ranges = [(0, 10), (10, 30)]\n"
            "class M:\n    arena = A()\n"
            "pre = parallel.preflight(M(), 'cpu')\nassert pre['ranks'] == 1 and pre['backend'] == 'gloo' and pre['tower_ranges_checked'] == 2\n"
            "parallel.barrier()\nprint('ok')\n")
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, SVLA_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
    env.pop("SVLA_FORCE_DIST")
    r = subprocess.run([sys.executable, "-c", "from safevla_amd import parallel\nprint(parallel.init_from_env(), parallel.is_dist())"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "(0, 0, 1) False" in r.stdout, r.stdout + r.stderr[-2000:]
