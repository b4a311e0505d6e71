#!/usr/bin/env python3
"""bench.py -- env-steps/s through the PPO-Lagrangian update on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = ONE full PPO-Lagrangian update over a synthetic rollout already resident in HBM:
  reward+cost GAE  ->  lambda update  ->  update_repeats(4) x [3 towers x (forward, fused loss, backward),
  all-reduce of the flat gradient arena, global-norm clip + Adam].
Workload at N=1 = BASELINE.json configs[2] (C3, the largest single-GPU configuration): PickUp, 64 envs x 256 steps, cost
constraint active (cost_limit 2.31964, lambda moves), 12 goal tokens, all 16 384 rows in one pass per tower (--env-chunk N accumulates).
N > 1: the same 64 envs x 256 steps on EVERY GPU (weak scaling; C4 = "256 envs over 8 GPUs" is the 32-envs/GPU point and is
reported, with C2 and the 64-token probe, under "secondary" at N=1).  `--gpus N` without a torchrun environment re-executes
itself under torch.distributed.run (one rank per GPU, RCCL) and fails loudly if fewer than N GPUs are visible.
Prints ONE JSON line (rank 0) with the roofline of the dominant kernel (the bf16 MFMA GEMM, timed with HIP events on
its launch stream in a separate instrumented update) and a CPU baseline (the fp32 oracle port on the host cores, bounded
sample).  The oracle is only the checker/baseline leg here -- the measured path is the HIP library.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_TBPS = 8.0           # HBM3E peak (MI355X_MICROARCH.md)


def flops_per_update(R, S, L, U, epochs):
    """Algorithmic FLOPs (2*MAC) of the trainable path, SURVEY.md section 8(d): per row per tower forward,
    x3 (fwd + 2x bwd) x 3 towers x epochs.  Frozen T5 (once per unique goal) excluded."""
    d = 512
    fusion = 3 * S * (4 * d * d + 2 * d * 2048 + 2 * S * d) * 2
    compress = 2 * 84 * (384 * d + d * d + d * d) * 2
    text = L * d * d * 2
    decoder = 3 * (4 * d * d + 3 * d * 1536 + 2 * 256 * d) * 2 + d * d * 2 + 21 * d * 2
    return R * 3 * 3 * epochs * (fusion + compress + decoder) + U * 3 * 3 * epochs * text


class GemmTimer:
    """HIP-event timing of every MFMA-kernel launch (on torch's current stream = the launch stream) during one update."""

    def __init__(self, ops):
        self.ops, self.rec = ops, {"gemm_nt256": [], "gemm_nt": [], "gemm_tn": [], "attn_fwd": [], "attn_bwd": []}
        self.orig = {"gemm_nt": ops.gemm_nt, "gemm_tn": ops.gemm_tn_acc, "attn_fwd": ops.attn_fwd, "attn_bwd": ops.attn_bwd}

    def _wrap(self, key, fn, flops):
        def wrapped(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **kw)
            e1.record()
            k2 = key
            if key == "gemm_nt":   # same dispatch rule as svla_gemm_nt_bf16: big row-streaming shapes run the persistent 256x256 kernel
                M, N, K = a[2], a[3], a[4]
                if (not kw.get("out_f32") and (N % 256 == 0 or (N % 128 == 0 and N >= 384)) and K % 64 == 0 and K >= 128
                        and ((M + 255) // 256) * ((N + 255) // 256) >= 160):
                    k2 = "gemm_nt256"
            self.rec[k2].append((e0, e1, flops(*a, **kw)))
            return out

        return wrapped

    def __enter__(self):
        o = self.ops
        o.gemm_nt = self._wrap("gemm_nt", self.orig["gemm_nt"], lambda A, B, M, N, K, **kw: 2.0 * M * N * K)
        o.gemm_tn_acc = self._wrap("gemm_tn", self.orig["gemm_tn"], lambda dY, X, dW, M, N, K, **kw: 2.0 * M * N * K)
        o.attn_fwd = self._wrap("attn_fwd", self.orig["attn_fwd"],
                                lambda q, k, v, ld, rows, S, H, scale, **kw: 4.0 * (kw.get("Sq") or S) * S * 64 * rows * H)
        o.attn_bwd = self._wrap("attn_bwd", self.orig["attn_bwd"],
                                lambda q, k, v, ld, o_, ldo, lse, do, lddo, dq, dk, dv, ldd, rows, S, H, scale, **kw:
                                10.0 * (kw.get("Sq") or S) * S * 64 * rows * H)
        return self

    def __exit__(self, *a):
        o = self.ops
        o.gemm_nt, o.gemm_tn_acc, o.attn_fwd, o.attn_bwd = (self.orig[k] for k in ("gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd"))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for key, rec in self.rec.items():
            ms = [e0.elapsed_time(e1) for e0, e1, _ in rec]
            fl = sum(f for _, _, f in rec)
            tot = sum(ms) * 1e-3
            out[key] = dict(launches=len(ms), avg_ms=sum(ms) / max(1, len(ms)), tflops=fl / max(tot, 1e-12) / 1e12, flops=fl, total_s=tot)
        return out


def acting_bench(model, st, B, dev, n=24):
    """Secondary number ("report separately the rollout-time throughput"): the acting path that precedes the update --
    frozen DINOv2 ViT on 2 uint8 frames per env step + single-step 3-tower forward with the llama KV caches."""
    from safevla_amd.preproc import DinoViTPreprocessor

    step_in = lambda t: ({k: v[t:t + 1] for k, v in st.observations.items()}, None, st.prev_actions[t:t + 1], st.masks[t:t + 1])

    def policy_rate(mode):
        for t in model.towers:
            t.time_step_counter, t._kv = 0, None
        if mode is None:
            model.enable_acting_graphs(False)
        else:
            model.enable_acting_graphs(True, backend=mode)
        with torch.no_grad():
            for t in range(4):
                model(*step_in(t))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(4, 4 + n):
                model(*step_in(t))
            torch.cuda.synchronize()
            r = n * B / (time.perf_counter() - t0)
        model.enable_acting_graphs(False)
        for t in model.towers:
            t.time_step_counter, t._kv = 0, None
        return r

    pol_eager, pol_plan, pol = policy_rate(None), policy_rate("plan"), policy_rate("hipgraph")
    grouped = bool(getattr(model, "grouped_towers", False))
    model.grouped_towers = False            # the same recorded step replayed as three call lists on three streams (the pre-round-6 acting path): A/B of the tower-grouped launches
    model.invalidate_recorded()
    pol_plan_3s = policy_rate("plan")
    model.grouped_towers = grouped
    model.invalidate_recorded()
    model.enable_acting_plans(True)         # back to the default acting path
    vit = DinoViTPreprocessor("rgb_raw", "rgb_dinov2", device=dev)
    fr = torch.randint(0, 256, (2 * B, 224, 384, 3), device=dev, dtype=torch.uint8)
    vit.process({"rgb_raw": fr})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        vit.process({"rgb_raw": fr})
    torch.cuda.synchronize()
    fps = 3 * 2 * B / (time.perf_counter() - t0)
    best = max(pol, pol_eager, pol_plan)
    return {"policy_single_step_env_steps_per_s": round(pol_plan, 1), "policy_single_step_three_streams_env_steps_per_s": round(pol_plan_3s, 1),
            "policy_single_step_ms": round(1e3 * B / pol_plan, 3), "tower_grouped_launches": grouped,
            "policy_single_step_eager_env_steps_per_s": round(pol_eager, 1),
            "policy_single_step_hipgraph_env_steps_per_s": round(pol, 1),
            "vit_frames_per_s": round(fps, 1),
            "acting_env_steps_per_s": round(1.0 / (1.0 / best + 2.0 / fps), 1), "envs": B,
            "note": "per env step: 2 frames (224x384) through DINOv2 ViT-S/14 + one KV-cached 3-tower step; eager = one Python-issued launch "
                    "per kernel, hipgraph = the same step captured once and replayed (model.enable_acting_graphs); policy_single_step = the recorded step of the three towers replayed as "
                    "tower-grouped launches (one grid per kernel, blockIdx.z = tower: csrc/launch.h) behind one staging launch, three_streams = the same recorded step as three call lists on three streams; synthetic frames"}


def north_star_probe(model, dev, rows_T=32, rows_B=8, L=64):
    """Secondary number named by BASELINE.json's north_star: MFMA-roofline fraction of the policy forward+backward at batch 256
    with a 64-token instruction (3 towers, fused losses, no optimiser step; pre-encoded features -- the frozen ViT is not part of
    the reference's update path).  Algorithmic FLOPs per SURVEY 8(d) / measured time / 2.5 PFLOP/s."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=rows_T, B=rows_B, L=L, task="Fetch", seed=99), device=dev)
    eng = PPOLagEngine(model, PPOLagConfig())
    st.compute_returns(nxt["next_value"], nxt["next_c_value"])
    batch = st.batch_slice(0, rows_B)
    R = rows_T * rows_B

    def once():
        model.zero_grad()
        eng._accumulate(batch, R, 0.1, cache_key="probe")     # like the epochs of one update: the first pass records, the others replay

    from safevla_amd import ops
    with GemmTimer(ops) as gt:      # the recording pass: every MFMA launch goes through the Python bindings once -> the FLOPs this engine EXECUTES
        once()
    executed = sum(v["flops"] for v in gt.summary().values())
    torch.cuda.synchronize()
    once()                      # (one untimed replay: the first pass after the recording still allocates)
    torch.cuda.synchronize()
    reps = []
    for _ in range(3):          # best of three groups of four: a single allocator / clock hiccup used to move a three-iteration mean by 30 %
        t0 = time.perf_counter()
        for _ in range(4):
            once()
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / 4 * 1e3)
    ms = min(reps)
    U = int(st.observations["goal_token_ids"][:rows_T].reshape(R, -1).unique(dim=0).shape[0])
    fl = flops_per_update(R, 169 + L, L, U, 1)
    # the shape as north_star writes it -- "batch 256 x (2 x 3 x 224 x 224 + 64 tok)" -- includes the image encoder on two 224 x 224 frames per row: the
    # frozen DINOv2 ViT-S/14 on 512 uint8 frames (16 x 16 patches + class token) in front of the same 3-tower forward + backward
    as_written = None
    try:
        from safevla_amd.preproc import DINO_RGB_MEANS, DINO_RGB_STDS, DinoViTPreprocessor
        vit = DinoViTPreprocessor("rgb_raw", "rgb_dinov2", device=dev).vit
        fr = torch.randint(0, 256, (2 * R, 224, 224, 3), device=dev, dtype=torch.uint8)
        with GemmTimer(ops) as gv:
            vit.patch_tokens(fr, DINO_RGB_MEANS, DINO_RGB_STDS, crop_x=0)
        vit_fl = sum(v["flops"] for v in gv.summary().values())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            vit.patch_tokens(fr, DINO_RGB_MEANS, DINO_RGB_STDS, crop_x=0)
        torch.cuda.synchronize()
        vit_ms = (time.perf_counter() - t0) / 3 * 1e3
        as_written = {"vit_frames": 2 * R, "vit_ms": round(vit_ms, 2), "vit_executed_tflop": round(vit_fl / 1e12, 2), "vit_tflops": round(vit_fl / (vit_ms * 1e-3) / 1e12, 1),
                      "combined_ms": round(vit_ms + ms, 2), "combined_executed_tflop": round((vit_fl + executed) / 1e12, 2),
                      "combined_frac_of_bf16_mfma_peak": round((vit_fl + executed) / ((vit_ms + ms) * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                      "note": "frozen DINOv2 ViT-S/14 forward on 2 x 256 frames of 224 x 224 (uint8 -> normalise -> 14 x 14 patches -> 12 blocks, S = 257) + the 3-tower "
                              "forward + backward above; executed MFMA FLOPs / (ViT time + policy time) / 2.5 PFLOP/s"}
        del fr, vit
        torch.cuda.empty_cache()
    except Exception as e:
        as_written = {"error": repr(e)[:200]}
    return {"rows": R, "goal_tokens": L, "ms_fwd_bwd_3_towers": round(ms, 2), "ms_groups_of_four": [round(x, 2) for x in reps], "as_written_with_image_encoder": as_written,
            "executed_mfma_tflop": round(executed / 1e12, 2), "executed_tflops": round(executed / (ms * 1e-3) / 1e12, 1),
            "frac_of_bf16_mfma_peak": round(executed / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "reference_schedule_tflop": round(fl / 1e12, 2), "reference_schedule_frac_of_peak": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "note": "north_star secondary target (>= 0.70) -- small batch: 256 rows x 233 tokens.  frac_of_bf16_mfma_peak divides the MFMA FLOPs the engine "
                    "EXECUTES (GEMMs + attention, counted launch by launch) by the time; reference_schedule_* divides SURVEY 8(d)'s FLOPs of the reference's "
                    "schedule (which runs the whole last fusion layer and T5 per row) -- VERDICT r4"}


def _oracle_epochs(T, B, L, train_mode, device="cpu", epochs=1, autocast=False):
    """`epochs` x [3-tower forward, SafePPOLogGrad + SafePPOValue, backward, clip 0.5, Adam] of the fp32 torch restatement (oracle port) on
    `device` with stock PyTorch ops; returns seconds.  The checker / baseline leg only -- never the measured product path."""
    from oracle import ref_loss, ref_model

    torch.manual_seed(0)

    class _Tok:  # fixed-length synthetic ids, same as the GPU workload
        def __call__(self, goals, return_tensors="pt", padding=True):
            ids = torch.randint(3, 32000, (len(goals), L))
            return {"input_ids": ids, "attention_mask": torch.ones_like(ids)}

    m = ref_model.RefSafeActorCritic(_Tok(), max_steps=500, max_batch=B, dropout=0.1 if train_mode else 0.0).to(device)
    m.train(train_mode)       # the same mode as the GPU leg (the reference trains with dropout 0.1 on); T5 stays deterministic
    for t in (m, m.critic_tsfm, m.c_critic_tsfm):
        t.visual_encoder.text_encoder.eval()
    params = [p for n_, p in m.named_parameters() if "text_encoder" not in n_]
    opt = torch.optim.Adam(params, lr=2e-5)
    dv = lambda x: x.to(device)
    obs = {"rgb_dinov2": dv(torch.randn(T, B, 384, 7, 12)), "manipulation_rgb_dinov2": dv(torch.randn(T, B, 384, 7, 12)),
           "natural_language_spec": dv(torch.zeros(T, B, 1000, dtype=torch.uint8)), "time_step": dv(torch.arange(T)[:, None].expand(T, B).contiguous()),
           "traj_index": dv(torch.zeros(T, B, dtype=torch.int64)), "an_object_is_in_hand": dv(torch.zeros(T, B, 1, dtype=torch.int64))}
    batch = {"actions": torch.randint(0, 20, (T, B)), "old_action_log_probs": torch.full((T, B), -3.0), "adv_targ": torch.randn(T, B, 1),
             "c_adv_targ": torch.randn(T, B, 1), "returns": torch.randn(T, B, 1), "values": torch.randn(T, B, 1), "c_returns": torch.randn(T, B, 1)}
    batch = {k: dv(v) for k, v in batch.items()}
    pa, mk = dv(torch.randint(0, 20, (T, B))), dv(torch.ones(T, B, 1))
    is_gpu = str(device) != "cpu"

    def epoch():
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast and is_gpu):
            out, _ = m(obs, None, pa, mk)
        total, _ = ref_loss.safe_ppo_log_grad(out["logits"].float(), out["values"].float(), batch, 0.1)
        (total + ref_loss.safe_ppo_value(out["c_values"].float(), batch["c_returns"])).backward()
        torch.nn.utils.clip_grad_norm_(params, 0.5)
        opt.step()

    if is_gpu:               # one untimed epoch: rocBLAS / MIOpen handle creation and kernel selection
        epoch()
        torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(epochs):
        epoch()
    if is_gpu:
        torch.cuda.synchronize()
    return time.time() - t0


def cpu_baseline(T=32, B=8, L=12, train_mode=True, threads=None):
    """fp32 CPU port (oracle) of the same update on a bounded sample: one epoch = 3-tower forward, SafePPOLogGrad +
    SafePPOValue, backward, clip 0.5, Adam; env-steps/s = T*B / (4 epochs)."""
    n = threads or min(16, os.cpu_count() or 1)   # more threads only add fork/join overhead on these op sizes (measured: 256 threads 40x slower)
    torch.set_num_threads(n)
    dt = _oracle_epochs(T, B, L, train_mode)
    return {"value": T * B / (4.0 * dt), "unit": "env-steps/s", "cores": n, "kind": "port",
            "sample": f"fp32 torch-CPU oracle, 1 of 4 epochs timed on T={T} x B={B} rows ({dt:.1f} s), L={L}, " + ("train mode (dropout 0.1)" if train_mode else "eval mode") + ", scaled x4"}


def cpu_c1_full(train_mode=True):
    """BASELINE configs[0] (C1: 4 envs x 32 steps) timed IN FULL on the host CPU: all 4 epochs of the update."""
    n = min(16, os.cpu_count() or 1)
    torch.set_num_threads(n)
    dt = _oracle_epochs(32, 4, 4, train_mode, epochs=4)
    return {"value": 32 * 4 / dt, "unit": "env-steps/s", "cores": n, "kind": "port", "seconds_per_update": round(dt, 1),
            "sample": "C1 in full: 4 envs x 32 steps, L=4, 4 epochs x 3 towers, fp32 torch-CPU oracle, " + ("train mode" if train_mode else "eval mode")}


def stock_rocm_baseline(dev, L=12, train_mode=True, T=64, B=8):
    """SURVEY 8(d) / BASELINE.md 2.3 intermediate baseline: the same restatement on the MI355X through stock PyTorch-ROCm ops (rocBLAS /
    MIOpen / eager elementwise kernels; no custom HIP kernel), fp32 as the reference runs it and under bf16 autocast."""
    out = {"sample": f"oracle port on cuda, 1 of 4 epochs timed on T={T} x B={B} rows after one warm-up epoch, L={L}, scaled x4", "unit": "env-steps/s"}
    for name, ac in (("fp32", False), ("bf16_autocast", True)):
        dt = _oracle_epochs(T, B, L, train_mode, device=dev, autocast=ac)
        out[name] = round(T * B / (4.0 * dt), 1)
    torch.cuda.empty_cache()
    return out


def kernel_sources_sha():
    """sha256 over the HIP sources + headers the library is built from (what a profiles/*_pmc_hbm_traffic.json must have been measured on)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "safevla_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "safevla_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "safevla_amd", "asmgen", "*.py")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def live_traffic(M, N, K, flavour, kernel_substr, timeout=180):
    """HBM bytes per launch of ONE GEMM shape measured in THIS run (the stored-profile figure is a claim about another run): tools/one_gemm.py under two
    rocprofv3 passes -- --kernel-trace --pmc FETCH_SIZE, then WRITE_SIZE, separately, as MI355X_MICROARCH.md prescribes -- in child processes; counters are KiB and
    FETCH_SIZE reports half of a wide coalesced stream on gfx950 (the same corrections as tools/prof_summarize.py).  Returns a dict or {"error": ...}; never raises."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    out, tmp = {}, tempfile.mkdtemp(prefix="svla_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "x", "--", sys.executable, os.path.join(ROOT, "tools", "one_gemm.py"),
                   str(M), str(N), str(K), "asm", "4", flavour]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return {"error": f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {r.stderr[-200:]}"}
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, value, duration from counters_collection where counter_name=?", (counter,)).fetchall()
            rows = [(v, du) for n_, v, du in rows if kernel_substr in n_]
            if not rows:
                return {"error": f"no {kernel_substr} dispatch in the {counter} pass"}
            out[counter] = sum(v for v, _ in rows) / len(rows) * 1024.0
            out["dur_us"] = sum(du for _, du in rows) / len(rows) / 1e3
        fetch, write = 2.0 * out["FETCH_SIZE"], out["WRITE_SIZE"]
        return {"kernel": kernel_substr, "M": M, "N": N, "K": K, "flavour": flavour, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                "hbm_bytes_per_launch": fetch + write, "avg_duration_us_under_pmc": round(out["dur_us"], 1),
                "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two child-process passes of tools/one_gemm.py in this bench run); bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024"}
    except Exception as e:
        return {"error": repr(e)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _stamped_profile(suffix):
    """newest profiles/*<suffix> measured on THIS build's kernel sources (hash-stamped), or (None, None)"""
    for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix)), reverse=True):
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", cand)))
            if doc.get("kernel_sources_sha256") == kernel_sources_sha():
                return doc, cand
        except Exception:
            pass
    return None, None


def family_roofs(allk, ms):
    """Both roofs per kernel family (d = 512 sits at the ridge, 165-410 FLOP/B against 312).  MFMA side: live HIP-event FLOP/s of this
    run.  HBM side: the family's FLOP-per-HBM-byte from the rocprofv3 counter passes of the same command on the same kernel sources (profiles/, hash-stamped;
    the profiled run holds one update + the rollout's forward pass, so the intensity -- not the absolute byte count -- is what carries over), applied to the
    live FLOP/s; attention / LayerNorm / whole-run figures are the profile's own bytes over the profile's own durations."""
    shp, src_s = _stamped_profile("_pmc_hbm_traffic_by_shape.json")
    ker, src_k = _stamped_profile("_pmc_hbm_traffic.json")
    out = {"_sources": {"by_shape": src_s, "by_kernel": src_k}}
    for fam, key, pick in (("nt_gemms", "gemm_nt256", lambda k: "tn" not in k), ("tn_gemms", "gemm_tn", lambda k: "tn" in k)):
        v = allk.get(key)
        if not v or not v["launches"]:
            continue
        row = {"achieved_tflops": round(v["tflops"], 1), "mfma_frac": round(v["tflops"] / PEAK_BF16_TFLOPS, 4), "share_of_update": round(v["total_s"] / (ms * 1e-3), 3)}
        if shp is not None:
            rows = [r for r in shp["shapes"] if "error" not in r and pick(r["kernel"]) and r["M"] >= 65536]
            fl = sum(2.0 * r["M"] * r["N"] * r["K"] * r["launches"] for r in rows)
            by = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows)
            alg = sum(r["algorithmic_bytes_per_launch"] * r["launches"] for r in rows)
            if by > 0:
                row.update(flop_per_hbm_byte=round(fl / by, 1), hbm_TBps=round(v["tflops"] / (fl / by), 3), hbm_frac=round(v["tflops"] / (fl / by) / PEAK_HBM_TBPS, 4),
                           traffic_over_algorithmic=round(by / alg, 3))
        out[fam] = row
    for fam, key in (("attn_fwd", "attn_fwd"), ("attn_bwd", "attn_bwd")):
        v = allk.get(key)
        if v and v["launches"]:
            out[fam] = {"achieved_tflops": round(v["tflops"], 1), "mfma_frac": round(v["tflops"] / PEAK_BF16_TFLOPS, 4), "share_of_update": round(v["total_s"] / (ms * 1e-3), 3),
                        "avg_launch_ms": round(v["avg_ms"], 4)}
    if ker is not None:
        tot_b = tot_t = 0.0
        for name, r in ker["kernels"].items():
            tot_b += r["hbm_bytes_per_launch"] * r["launches"]
            tot_t += r["avg_duration_us_profiled"] * 1e-6 * r["launches"]
            for fam, pat in (("attn_fwd", "attn_fwd_persist"), ("attn_bwd", "attn_bwd_fused"), ("norm_fwd", "norm_fwd_kernel"), ("norm_bwd", "norm_bwd_kernel")):
                if pat in name and r["launches"] >= 8 and r["avg_duration_us_profiled"] > 200:
                    tb = r["hbm_bytes_per_launch"] / (r["avg_duration_us_profiled"] * 1e-6) / 1e12
                    out.setdefault(fam, {}).update(hbm_TBps_profiled=round(tb, 3), hbm_frac=round(tb / PEAK_HBM_TBPS, 4), hbm_GB_per_launch=round(r["hbm_bytes_per_launch"] / 1e9, 3))
        if tot_t > 0:
            out["whole_profiled_run"] = {"hbm_TB": round(tot_b / 1e12, 2), "kernel_s": round(tot_t, 3), "hbm_TBps": round(tot_b / tot_t / 1e12, 3), "hbm_frac": round(tot_b / tot_t / 1e12 / PEAK_HBM_TBPS, 4)}
    return out


def parity_gate(dev, T=8, B=4):
    """the parity gate of the same run.  One seeded (T x B)-row minibatch through the CPU oracle (the checker) and through
    the HIP path with the same weights, eval mode: the fp32 verification mode must agree at fp32 tolerance (1e-4), the bf16 product path
    on its documented ladder (3e-2).  Reports relative-to-max errors of logits / values / c_values and the SafePPOLogGrad scalars."""
    import numpy as np

    from oracle import ref_loss, ref_model
    from safevla_amd.losses import SafePPOLogGrad
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.text import GoalTokenizer, str_to_bytes

    rs = np.random.RandomState(7)
    goals = ["find a mug", "pick up the bowl", "fetch a red apple", "find a mug"]
    obs = {"rgb_dinov2": torch.from_numpy(rs.standard_normal((T, B, 384, 7, 12)).astype(np.float32)),
           "manipulation_rgb_dinov2": torch.from_numpy(rs.standard_normal((T, B, 384, 7, 12)).astype(np.float32)),
           "natural_language_spec": torch.from_numpy(np.stack([np.stack([str_to_bytes(goals[b]).reshape(-1) for b in range(B)]) for _ in range(T)])),
           "time_step": torch.arange(T)[:, None].expand(T, B).contiguous(), "traj_index": torch.from_numpy((np.arange(T)[:, None] >= 5).astype(np.int64) + np.zeros((T, B), np.int64)),
           "an_object_is_in_hand": torch.from_numpy(rs.randint(0, 2, (T, B, 1)))}
    obs["time_step"] = torch.where(obs["traj_index"] > 0, obs["time_step"] - 5, obs["time_step"])          # an episode boundary at step 5
    pa, mk = torch.from_numpy(rs.randint(0, 20, (T, B))), torch.ones(T, B, 1)
    mk[5] = 0
    batch = {"actions": torch.from_numpy(rs.randint(0, 20, (T, B))), "old_action_log_probs": torch.full((T, B), -3.0), "adv_targ": torch.randn(T, B, 1),
             "c_adv_targ": torch.randn(T, B, 1), "returns": torch.randn(T, B, 1), "values": torch.randn(T, B, 1)}
    d = lambda x: {k: v.to(dev) for k, v in x.items()}
    rel = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    out = {"sample": f"T={T} x B={B} rows, 3 distinct goals, one episode boundary, eval mode, lambda 0.37; relative-to-max errors vs the fp32 CPU oracle"}
    ref = None
    for prec in ("fp32", "bf16"):
        torch.manual_seed(11)
        m = SafeDinoLLAMATxNavActorCriticSeparate(device=dev, precision=prec).eval()
        if ref is None:
            ref = ref_model.RefSafeActorCritic(GoalTokenizer(), max_batch=B).eval()
            ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
            with torch.no_grad():
                ro, _ = ref(obs, None, pa, mk)
            _, ri = ref_loss.safe_ppo_log_grad(ro["logits"], ro["values"], batch, 0.37)
            sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        else:
            m.load_state_dict(sd)
        with torch.no_grad():
            aco, _ = m(d(obs), None, pa.to(dev), mk.to(dev))
        _, info = SafePPOLogGrad(0.1, 0.5, 0.0, use_clipped_value_loss=False, normalize_advantage=False).loss(0, d(batch), aco, lagrangian_multiplier=torch.tensor(0.37))
        e = {"logits": rel(aco.distributions.logits.float().cpu(), ref_loss.categorical(ro["logits"])), "values": rel(aco.values.cpu(), ro["values"]),
             "c_values": rel(aco.c_values.cpu(), ro["c_values"]),
             "loss_scalars": max(abs(info[k] - ri[k]) / max(1.0, abs(ri[k])) for k in ("ppo_total", "value", "action", "entropy"))}
        tol = 1e-4 if prec == "fp32" else 3e-2
        out[prec] = {k: float(f"{v:.3g}") for k, v in e.items()}
        out[prec].update(tolerance=tol, passed=bool(max(e.values()) < tol))
        del m, aco
        torch.cuda.empty_cache()
    return out


def _self_spawn(args):
    """`python bench.py --gpus N` outside torchrun: re-exec under torch.distributed.run, one rank per GPU (RCCL).  Never
    falls back to fewer ranks: a box with < N GPUs is an error."""
    import socket
    import subprocess

    n_vis = torch.cuda.device_count()
    if n_vis < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_vis} GPU(s) visible -- refusing to run fewer ranks than requested")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SVLA_BENCH_SPAWNED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def timed_updates(eng, st, nxt, ep, steps, warmup, world, dev):
    """W untimed + K timed full updates, bracketed by barrier + synchronize, MAX over ranks.  Returns (ms per update, last info)."""
    from safevla_amd import parallel

    def step():
        return eng.update(st, nxt["next_value"], nxt["next_c_value"], ep["episode_cost_sum"], ep["n_episodes"])

    info = None
    for _ in range(warmup):
        info = step()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        info = step()
    parallel.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timed_updates.per_rank_ms = [x / steps * 1e3 for x in parallel.gather_floats(dt, dev)]      # every rank's own clock over the same K steps
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    return dt / steps * 1e3, info, step


def secondary_config(model, dev, name, task, T, B, L, env_chunk, cost_limit, fp8=False, t5_per_row=False):
    """Another BASELINE configuration through the same engine (1 warm-up + 1 timed update; parity-test cases, not the bench line).
    fp8: the fusion-encoder attention on the e4m3 / e5m2 kernels (BASELINE configs[4]: "fp8 MFMA attention"), acting pass included."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    model.set_fp8_attention(fp8)
    model.t5_dropout_per_row = t5_per_row      # the reference's train-mode T5 statistics: one dropout realisation per (t, b) row and tower
    try:
        st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=L, task=task, seed=4321), device=dev)
        eng = PPOLagEngine(model, PPOLagConfig(env_chunk=env_chunk, cost_limit=cost_limit))
        ms, info, _ = timed_updates(eng, st, nxt, ep, 1, 1, 1, dev)
    finally:
        model.set_fp8_attention(False)
        model.t5_dropout_per_row = False
    S = 169 + L
    U = int(st.observations["goal_token_ids"][:T].reshape(T * B, -1).unique(dim=0).shape[0])
    return {"workload": name, "task": task, "rollout_steps": T, "envs": B, "goal_tokens": L, "env_chunk": env_chunk, "attention": "fp8 (e4m3 / e5m2)" if fp8 else "bf16",
            "t5_dropout": "per (t, b) row and tower, as the reference draws it (allenact_dino_transformer.py:193,591-605)" if t5_per_row else "per unique goal and pass (default)",
            "ms_per_update": round(ms, 2),
            "env_steps_per_s": round(T * B / (ms * 1e-3), 1),
            "reference_equivalent_tflops": round(flops_per_update(T * B, S, L, U, 4) / (ms * 1e-3) / 1e12, 1),
            "lagrangian_multiplier": round(info["lagrangian_multiplier"], 6), "Jc": round(info["Jc"], 4)}


class _EnvWithImageEncoder:
    """SynthVectorEnv whose observations carry what the frozen image encoder makes of two uint8 camera frames per env (a fixed synthetic frame batch: the content
    is irrelevant to the timing), the way the reference's sensor preprocessors run inside the rollout (dino_preprocessors.py:38-125)."""

    def __init__(self, env, vit, frames):
        self.env, self.vit, self.frames, self.B = env, vit, frames, env.B

    def _encode(self, obs):
        self.vit.process_tokens_all_cameras(self.frames, obs["dino_tokens"])      # both cameras of all envs in one pass of the frozen trunk
        return obs

    def reset(self):
        return self._encode(self.env.reset())

    def step(self, actions, want_results=False):
        obs, reward, cost, done, res = self.env.step(actions, want_results)
        return self._encode(obs), reward, cost, done, res

    def pop_episode_costs(self):
        return self.env.pop_episode_costs()


def collect_then_update(model, dev, T=256, B=64, L=12, task="PickUp", cost_limit=2.31964, iters=2):
    """The whole training iteration as training/online/dinov2_vits_tsfm_base.py runs it (safevla_amd/train.py): collect T steps from B synthetic environments through
    the ACTING path (KV-cached single-step 3-tower forward, action sampling, env.step, storage.add), then one full PPO-Lagrangian update on that rollout.  Two variants:
    pre-encoded DINOv2 features from the environment, and two uint8 frames per env and step through the frozen ViT-S/14 inside the rollout (what the reference's sensor
    preprocessors do).  One untimed iteration, then `iters` timed ones; env-steps/s of the loop and its split."""
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.preproc import DinoViTPreprocessor
    from safevla_amd.storage import RolloutStorage
    from safevla_amd.synth_env import SynthVectorEnv, collect_rollout

    out = {"workload": f"{task}, {B} envs x {T} steps collected through the acting path (synthetic vector env) + one full 3-tower update, {iters} timed iterations"}
    for variant in ("pre_encoded_features", "with_image_encoder"):
        env = SynthVectorEnv(B, L=L, task=task, seed=99, max_steps=500, device=dev)
        if variant == "with_image_encoder":
            vit = DinoViTPreprocessor("rgb_raw", "rgb_dinov2", device=dev)
            env = _EnvWithImageEncoder(env, vit, torch.randint(0, 256, (2 * B, 224, 384, 3), device=dev, dtype=torch.uint8))
        st = RolloutStorage(T, device=dev, store_tokens=True)
        st.initialize(env.reset(), num_samplers=B)
        eng = PPOLagEngine(model, PPOLagConfig(cost_limit=cost_limit))
        for t in model.towers:
            t.time_step_counter, t._kv = 0, None
        tc = tu = 0.0
        for it in range(iters + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nxt = collect_rollout(model, env, st, T)
            s_, n_ = env.pop_episode_costs()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng.update(st, nxt["next_value"], nxt["next_c_value"], s_, n_)
            st.after_updates()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if it > 0:
                tc += t1 - t0
                tu += t2 - t1
        out[variant] = {"env_steps_per_s": round(iters * T * B / (tc + tu), 1), "collect_s_per_iteration": round(tc / iters, 3), "update_s_per_iteration": round(tu / iters, 3),
                        "collect_ms_per_step_of_all_envs": round(tc / iters / T * 1e3, 3)}
        del eng, st, env
        torch.cuda.empty_cache()
    for t in model.towers:
        t.time_step_counter, t._kv = 0, None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--T", type=int, default=256)
    ap.add_argument("--envs-per-gpu", type=int, default=64)
    ap.add_argument("--L", type=int, default=12)
    ap.add_argument("--task", default="PickUp")
    ap.add_argument("--env-chunk", type=int, default=0, help="envs per gradient-accumulation chunk (0: the whole local minibatch in one pass -- 16 384 rows "
                    "keep ~140 GB of one tower's activations resident, which is what 288 GB of HBM are for; measured +1.9 %% over 2 chunks of 32)")
    ap.add_argument("--cost-limit", type=float, default=2.31964)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): every GPU runs --envs-per-gpu envs; strong: --global-envs envs (BASELINE configs[3]: Fetch, 256) are sharded "
                         "over the ranks with parallel.shard_envs, the reference's evenly_distribute_count_into_bins (training/online/base.py:208-224)")
    ap.add_argument("--global-envs", type=int, default=256)
    ap.add_argument("--deterministic", action="store_true", help="bitwise-repeatable gradients (64-bit fixed-point accumulation, PPOLagConfig.deterministic); "
                    "the default bench line is measured with the fp32 atomics")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc child processes that measure roofline.traffic_live in this run (~1 min)")
    ap.add_argument("--no-cpu-c1", action="store_true", help="skip the full CPU timing of BASELINE configs[0] inside cpu_baseline (~1 min of host time)")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--eval-mode", action="store_true", help="dropout off (the reference trains with the policy in train() mode: default here too)")
    ap.add_argument("--cpu-c1-full", action="store_true", help="only time BASELINE configs[0] (4 envs x 32 steps) IN FULL on the host CPU and exit")
    ap.add_argument("--fp8-attention", action="store_true", help="fusion-encoder attention on the e4m3 / e5m2 MFMA kernels in the HEADLINE run (BASELINE configs[4]: "
                    "--gpus 8 --scaling strong --global-envs 256 --task Mixed --L 64 --fp8-attention)")
    ap.add_argument("--t5-dropout-per-row", action="store_true", help="headline run with the reference's train-mode T5 statistics: one dropout realisation per (t, b) row and tower "
                    "(default: one per unique goal and pass; both figures are in the default line)")
    ap.add_argument("--grad-allreduce-bf16", action="store_true", help="the three per-tower gradient all-reduces cross xGMI as bf16 (126 MB instead of 252 MB per optimiser step)")
    args = ap.parse_args()

    if args.cpu_c1_full:
        print(json.dumps({"cpu_c1_full": cpu_c1_full(not args.eval_mode)}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_spawn(args)

    from safevla_amd import ops, parallel
    from safevla_amd.engine import PPOLagConfig, PPOLagEngine
    from safevla_amd.model import SafeDinoLLAMATxNavActorCriticSeparate
    from safevla_amd.synth_env import SynthSpec, fill_synthetic_rollout

    rank, local, world = parallel.init_from_env()
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per requested GPU")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(1234)          # identical frozen text encoder / initial weights on every rank ...
    model = SafeDinoLLAMATxNavActorCriticSeparate(device=dev)
    model.train(not args.eval_mode)
    if args.fp8_attention:
        model.set_fp8_attention(True)
    model.t5_dropout_per_row = bool(args.t5_dropout_per_row)
    if world > 1:                    # ... and made certain by a broadcast of every parameter and buffer
        parallel.broadcast_model_(model)
    # N > 1: the update's collectives on sentinels, checked on every rank BEFORE anything is timed (one async all-reduce per tower range of the
    # gradient arena, the cost reduction, the rank count) -- a mis-wired RCCL run fails here, not in the numbers
    pre = parallel.preflight(model, dev)
    if pre["ranks"] != max(1, args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus}: the collective pre-flight saw {pre['ranks']} rank(s)")
    torch.manual_seed(1234 + rank)   # per-rank streams from here on (sampling, synthetic environments)
    T, B = args.T, args.envs_per_gpu
    if args.scaling == "strong":      # fixed total work: this rank's share of the global env list (uneven shards are legal)
        _first, B = parallel.shard_envs(args.global_envs, world, rank)      # (first env, number of envs) of this rank
        if B <= 0:
            raise SystemExit(f"bench.py --scaling strong: {args.global_envs} envs cannot feed {world} ranks")
    chunk = args.env_chunk if 0 < args.env_chunk < B else None
    wire = "bf16" if args.grad_allreduce_bf16 else "fp32"
    cfg = PPOLagConfig(env_chunk=chunk, cost_limit=args.cost_limit, deterministic=args.deterministic, grad_allreduce_dtype=wire)
    eng = PPOLagEngine(model, cfg)
    st, nxt, ep = fill_synthetic_rollout(model, SynthSpec(T=T, B=B, L=args.L, task=args.task, seed=1234 + rank), device=dev)

    try:
        ms, info, step = timed_updates(eng, st, nxt, ep, args.steps, args.warmup, world, dev)
    except torch.OutOfMemoryError:
        # one pass over all local rows keeps ~8.5 MB of saved activations per row of ONE tower resident; if this GPU does not have that much
        # free memory, accumulate over env-chunks of 32 instead (exact in eval mode, equally valid noise in train mode) and say so in the line
        if world > 1:
            raise
        del eng
        torch.cuda.empty_cache()
        chunk = 32 if B > 32 else max(1, B // 2)
        cfg = PPOLagConfig(env_chunk=chunk, cost_limit=args.cost_limit, deterministic=args.deterministic, grad_allreduce_dtype=wire)
        eng = PPOLagEngine(model, cfg)
        ms, info, step = timed_updates(eng, st, nxt, ep, args.steps, args.warmup, world, dev)
    per_rank_ms = list(timed_updates.per_rank_ms)
    env_steps = T * (args.global_envs if args.scaling == "strong" else B * world)
    S, R = 169 + args.L, T * B
    U = int(st.observations["goal_token_ids"][:T].reshape(R, -1).unique(dim=0).shape[0])
    algo = flops_per_update(R, S, args.L, U, cfg.update_repeats)

    roof = None
    gt = None
    parity_failed = False
    if not args.no_roofline:
        # one extra, instrumented update: EVERY rank runs it (it contains the gradient / cost all-reduces), rank 0 times its
        # MFMA-kernel launches with HIP events
        if rank == 0:
            with GemmTimer(ops) as gt:
                step()
        else:
            step()
        parallel.barrier()
    if rank == 0 and gt is not None:
        allk = gt.summary()
        g = allk["gemm_nt256"]
        executed = sum(v["flops"] for v in allk.values())
        # HBM bytes per launch from the rocprofv3 PMC passes of this same command (tools/profile_round.sh -> profiles/).  The file carries the
        # hash of the kernel sources it was measured on: a file from another build is refused (traffic = null) instead of being quoted
        traffic, traffic_src, traffic_shape = None, None, None
        for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_hbm_traffic_by_shape.json")), reverse=True):
            try:
                doc = json.load(open(os.path.join(ROOT, "profiles", cand)))
                if doc.get("kernel_sources_sha256") != kernel_sources_sha():
                    continue
                # the (kernel, M, N, K) of the NT family with the largest share of the profiled update: ITS bytes per launch next to ITS algorithmic bytes
                nt = [r for r in doc["shapes"] if "error" not in r and ("nt_as" in r["kernel"] or "gemm_nt" in r["kernel"])]
                top = max(nt, key=lambda r: r["total_ms_profiled"])
                traffic, traffic_src = round(top["hbm_bytes_per_launch"]), cand
                traffic_shape = {k: top[k] for k in ("kernel", "M", "N", "K", "launches", "algorithmic_bytes_per_launch", "ratio_to_algorithmic", "avg_duration_us_profiled")}
                break
            except Exception:
                pass
        # the same quantity measured in THIS run, for the NT family's largest-share launch of the headline workload (linear1's forward: ReLU + dropout + sign bits,
        # N = 2048, K = 512 over all local rows): two rocprofv3 --pmc child processes, ~1 min; --no-live-traffic skips it
        traffic_live = None
        if world == 1 and not args.no_live_traffic:
            Ml = R * S
            traffic_live = live_traffic(Ml, 2048, 512, "relu_drop_bits", "svla_nt_as_f1d")
            if "error" not in traffic_live:
                alg = 2.0 * Ml * (512 + 2048) + Ml * 2048 / 8
                traffic_live.update(algorithmic_bytes_per_launch=alg, ratio_to_algorithmic=round(traffic_live["hbm_bytes_per_launch"] / alg, 3))
        families = None
        try:
            families = family_roofs(allk, ms)
        except Exception as e:
            families = {"error": repr(e)[:200]}
        traffic_stored = traffic
        if traffic_live and "error" not in traffic_live:
            traffic = round(traffic_live["hbm_bytes_per_launch"])      # the figure measured in THIS run takes precedence over the stored profile's
        roof = {"bound": "mfma", "kernel": "svla_gemm_nt_bf16, row-streaming shapes: svla_nt_as_* (generated gfx950 assembly, A panel stationary in 256 AGPRs per wave, K = 512) + "
                                           "gemm_nt8p_bf16_kernel (HIP, persistent 256x256x64 tile, K > 512 / residual epilogues); all launches of one update, HIP events", "achieved": round(g["tflops"], 1),
                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(g["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "traffic_unit": ("HBM bytes per launch of the family's largest-share (kernel, shape) of the headline workload, measured in THIS run: traffic_live (two rocprofv3 --pmc child passes); "
                                 f"traffic_stored_profile / traffic_shape = the same quantity from profiles/{traffic_src} (same kernel sources as this build)" if (traffic_live and "error" not in traffic_live) else
                                 f"HBM bytes per launch of the family's largest-share (kernel, shape) -- traffic_shape -- (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc, profiles/{traffic_src}, same kernel sources as this build)"
                                 if traffic is not None else "null: no profiles/*_pmc_hbm_traffic_by_shape.json was measured on this build's kernel sources (tools/profile_round.sh)"),
                "traffic_shape": traffic_shape, "traffic_live": traffic_live, "traffic_stored_profile": traffic_stored,
                "launches_per_update": g["launches"], "avg_launch_ms": round(g["avg_ms"], 4), "flops_per_launch": g["flops"] / max(1, g["launches"]),
                "share_of_update": round(g["total_s"] / (ms * 1e-3), 3),
                "other_mfma_kernels": {k: {"achieved_tflops": round(v["tflops"], 1), "launches": v["launches"], "avg_launch_ms": round(v["avg_ms"], 4),
                                           "share_of_update": round(v["total_s"] / (ms * 1e-3), 3)} for k, v in allk.items() if k != "gemm_nt256"},
                "families": families,
                "executed_mfma_tflop_per_update": round(executed / 1e12, 1),
                "executed_mfma_tflops_sustained": round(executed / (ms * 1e-3) / 1e12, 1),
                "executed_mfma_frac_of_peak": round(executed / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                # the other roof: this GEMM family is d = 512 wide (<= 256 FLOP/B at N = K = 512, below the 312 FLOP/B ridge of 2.5 PF / 8 TB/s),
                # so the HBM roof binds before the MFMA one; counter traffic per launch / live launch duration
                # (tiny configurations launch none of the 256-tile kernels: no launch duration to divide by, and the profiled traffic is not theirs)
                "hbm_view": None if traffic_stored is None else {"achieved_TBps": round(traffic_stored / (traffic_shape["avg_duration_us_profiled"] * 1e-6) / 1e12, 3), "peak_TBps": PEAK_HBM_TBPS,
                                                                 "frac": round(traffic_stored / (traffic_shape["avg_duration_us_profiled"] * 1e-6) / 1e12 / PEAK_HBM_TBPS, 4),
                                                                 "note": "traffic_shape's bytes over ITS profiled launch duration"},
                "hbm_view_live": None if not (traffic_live and "error" not in traffic_live) else {
                    "achieved_TBps": round(traffic_live["hbm_bytes_per_launch"] / (traffic_live["avg_duration_us_under_pmc"] * 1e-6) / 1e12, 3), "peak_TBps": PEAK_HBM_TBPS,
                    "frac": round(traffic_live["hbm_bytes_per_launch"] / (traffic_live["avg_duration_us_under_pmc"] * 1e-6) / 1e12 / PEAK_HBM_TBPS, 4),
                    "note": "traffic_live's bytes over its launch duration under the counter pass (the same launch at ~1.0 PFLOP/s: d = 512 GEMMs sit at the ridge)"}}
    cpu = None
    acting = None
    ns = None
    loop = None
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        # secondary measurements must never take the headline line down: each one is guarded on its own
        def guarded(fn, *a):
            try:
                return fn(*a)
            except Exception as e:
                torch.cuda.empty_cache()
                return {"error": repr(e)[:300]}

        del eng, step
        model.set_fp8_attention(False)         # the secondary legs choose their own attention / dropout variants
        model.t5_dropout_per_row = False
        acting = guarded(acting_bench, model, st, B, dev)
        del st, nxt
        torch.cuda.empty_cache()
        ns = guarded(north_star_probe, model, dev)
        loop = guarded(collect_then_update, model, dev)
        secondary = [guarded(secondary_config, model, dev, "C4-shard: one GPU's 32 envs of BASELINE configs[3] (Fetch, 256 envs over 8 GPUs)", "Fetch", 256, 32, 12, None, 2.31964),
                     guarded(secondary_config, model, dev, "C2: BASELINE configs[1] (ObjectNav, 32 envs x 128 steps)", "ObjectNav", 128, 32, 12, None, 2.31964),
                     guarded(secondary_config, model, dev, "C5-shard: mixed ObjectNav+PickUp+Fetch sampler (env e -> task e mod 3), 64-token instructions, 32 envs/GPU", "Mixed", 256, 32, 64, None, 2.31964),
                     guarded(secondary_config, model, dev, "C5-shard fp8: the same with the fusion-encoder attention on fp8 MFMA (BASELINE configs[4])", "Mixed", 256, 32, 64, None, 2.31964, True),
                     guarded(secondary_config, model, dev, "C3 with the reference-faithful T5 dropout (the headline workload, t5_dropout_per_row=True)", "PickUp", 256, 64, 12, None, 2.31964, False, True)]
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(L=args.L, train_mode=not args.eval_mode)
            one = cpu_baseline(T=8, B=2, L=args.L, train_mode=not args.eval_mode, threads=1)     # also at n = 1
            cpu["single_thread"] = {"value": one["value"], "unit": one["unit"], "sample": one["sample"]}
        except Exception as e:
            cpu = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": "failed: " + repr(e)[:200]}
        if not args.no_cpu_c1:
            try:         # BASELINE configs[0] (the reference's own CPU-runnable case) timed IN FULL on the host cores, in the same run (~1 min)
                cpu["c1_full"] = cpu_c1_full(not args.eval_mode)
            except Exception as e:
                cpu["c1_full"] = {"error": repr(e)[:200]}
        try:
            cpu["stock_pytorch_rocm"] = stock_rocm_baseline(dev, L=args.L, train_mode=not args.eval_mode)
        except Exception as e:          # the intermediate baseline must never take the bench line down
            cpu["stock_pytorch_rocm"] = {"error": repr(e)[:200]}
        try:
            cpu["parity_gate"] = parity_gate(dev)
        except Exception as e:
            cpu["parity_gate"] = {"error": repr(e)[:200]}
    if rank == 0:
        out = {"metric": "env-steps/sec through PPO-Lagrangian update", "value": round(env_steps / (ms * 1e-3), 1), "unit": "env-steps/s",
               "n_gpus": world, "rccl_ranks": (torch.distributed.get_world_size() if parallel.is_dist() else 1),
               "collective_preflight": pre, "ms_per_step_per_rank": {"min": round(min(per_rank_ms), 2), "max": round(max(per_rank_ms), 2)},
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
               "scaling": args.scaling,
               # a scaling CURVE needs this command at N = 1, 2, 4, 8 on a multi-GPU node (the driver's SCALE run): one line is one point, and only a line whose
               # ranks sat on distinct GPUs behind RCCL is a point of it (gloo ranks sharing one GPU are plumbing tests)
               "scaling_measured": bool(world > 1 and pre.get("backend") == "nccl" and torch.cuda.device_count() >= world),
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": ((f"C5: {args.task} task sampler, {args.global_envs} envs sharded over {world} GPU(s) ({B} on rank 0) x T={T}-step rollout, "
                                        f"fp8 MFMA attention (BASELINE configs[4], strong scaling), cost_limit {args.cost_limit}, "
                                        if (args.fp8_attention and args.task == "Mixed") else
                                        f"C4: {args.task}, {args.global_envs} envs sharded over {world} GPU(s) ({B} on rank 0) x T={T}-step rollout "
                                        f"(BASELINE configs[3], strong scaling), cost_limit {args.cost_limit}, ") if args.scaling == "strong" else
                                       f"C3: {args.task}, {B} envs/GPU x T={T}-step rollout, cost constraint active (cost_limit {args.cost_limit}) "
                                       f"(BASELINE configs[2]{'' if world == 1 else ', replicated per GPU: weak scaling'}), ") +
                                      f"L={args.L} goal tokens, S={S} fusion tokens, 3 towers x 4 epochs x 1 minibatch"
                                      f"{'' if chunk is None else f' in {B // chunk} env-chunks of {chunk}'}, Adam+clip",
                          "global_envs": args.global_envs if args.scaling == "strong" else B * world, "rollout_steps": T, "rows_per_gpu": R, "parallelism": f"dp{world}", "env_chunk": chunk,
                          "stage_losses": list(cfg.stage_losses), "weights": "random-init, reference geometry (168.9 M params)",
                          "dropout": 0.0 if args.eval_mode else 0.1, "deterministic_accumulation": bool(args.deterministic),
                          "attention": "fp8 (e4m3 / e5m2 MFMA)" if args.fp8_attention else "bf16", "t5_dropout": "per (t, b) row and tower (reference statistics)" if args.t5_dropout_per_row
                          else "per unique goal and pass", "grad_allreduce_dtype": wire},
               "reference_equivalent_tflop_per_update": round(algo / 1e12, 1),
               "note": "reference_equivalent counts SURVEY 8(d) FLOPs of the reference's schedule; the engine executes fewer (last fusion "
                       "layer only for the consumed token, T5 once per unique goal) -- see roofline.executed_mfma_*",
               "loss": {k: (round(v, 5) if isinstance(v, float) else v) for k, v in info.items()},
               "roofline": roof, "cpu_baseline": cpu, "acting": acting, "collect_then_update": loop, "north_star_batch256": ns, "secondary": secondary}
        for sc in (secondary or []):      # the reference-faithful noise statistics right beside the headline
            if isinstance(sc, dict) and "reference-faithful" in sc.get("workload", ""):
                out["value_with_reference_faithful_t5_dropout"] = {"value": sc.get("env_steps_per_s"), "ms_per_update": sc.get("ms_per_update"),
                                                                   "note": "the headline workload with one T5 dropout realisation per (t, b) row and tower, as the reference draws it; the headline draws one per unique goal and pass (SURVEY section 7 permits the de-duplication)"}
        pg = (cpu or {}).get("parity_gate")
        if pg is not None:      # the gate ran: a regression against the CPU oracle (or an exception inside the gate) fails the run
            out["parity_ok"] = bool("error" not in pg and pg.get("fp32", {}).get("passed") and pg.get("bf16", {}).get("passed"))
            parity_failed = not out["parity_ok"]
        print(json.dumps(out), flush=True)
    if parallel.is_dist():
        parallel.barrier()
        torch.distributed.destroy_process_group()
    if parity_failed:
        raise SystemExit("bench.py: parity gate FAILED against the CPU oracle (see cpu_baseline.parity_gate in the line above)")


if __name__ == "__main__":
    main()
