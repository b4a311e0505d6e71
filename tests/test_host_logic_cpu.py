"""Host logic that needs no GPU: Lagrange multiplier mirror vs the oracle, env sharding, tokenizer, API containers."""
import numpy as np
import pytest
import torch

from oracle.ref_rollout import RefLagrange
from safevla_amd import parallel
from safevla_amd.api import CategoricalDistr
from safevla_amd.lagrange import Lagrange
from safevla_amd.text import GoalTokenizer, bytes_to_str, str_to_bytes


def test_lagrange_matches_torch_adam_oracle():
    a, b = Lagrange(2.31964, 0.001, 0.035), RefLagrange(2.31964, 0.001, 0.035)
    rs = np.random.RandomState(0)
    for jc in rs.uniform(0.0, 6.0, 200):
        x, y = a.update_lagrange_multiplier(float(jc)), b.update(float(jc))
        assert abs(x - y) <= 1e-6 * max(1.0, abs(y)), (x, y)
    assert a.lagrangian_multiplier >= 0.0
    sd = a.state_dict()
    c = Lagrange(2.31964)
    c.load_state_dict(sd)
    assert c.update_lagrange_multiplier(3.0) == a.update_lagrange_multiplier(3.0)
    s = Lagrange(1.0, 0.5, 0.1, "SGD", lagrangian_upper_bound=0.55)
    assert abs(s.update_lagrange_multiplier(2.0) - 0.55) < 1e-7      # 0.5 + 0.1*1 = 0.6 -> clamped
    assert s.update_lagrange_multiplier(-100.0) == 0.0


def test_env_sharding_matches_reference_bins():
    assert parallel.evenly_distribute_count_into_bins(10, 4) == [3, 3, 2, 2]
    assert [parallel.shard_envs(256, 8, r) for r in range(8)] == [(32 * r, 32) for r in range(8)]
    cover = []
    for r in range(3):
        s, n = parallel.shard_envs(32, 3, r)
        cover += list(range(s, s + n))
    assert cover == list(range(32))


def test_goal_bytes_and_tokenizer():
    row = str_to_bytes("find a mug", 1000).reshape(-1)
    assert row.shape == (1000,) and bytes_to_str(row) == "find a mug"
    tok = GoalTokenizer()
    enc = tok(["find a mug", "navigate to the red apple"], return_tensors="pt", padding=True)
    assert enc["input_ids"].shape == (2, 6) and enc["attention_mask"].sum().item() == 4 + 6
    assert enc["input_ids"][0, 3].item() == 1 and enc["input_ids"][0, 4].item() == 0      # EOS then PAD
    assert tok.encode("Find A Mug") == tok.encode("find a mug")


def test_categorical_distr_matches_torch():
    lg = torch.randn(5, 3, 20)
    d, t = CategoricalDistr(lg), torch.distributions.Categorical(logits=lg)
    a = torch.randint(0, 20, (5, 3))
    assert torch.allclose(d.log_prob(a), t.log_prob(a), atol=1e-6) and torch.allclose(d.entropy(), t.entropy(), atol=1e-6)
    assert d.sample().shape == (5, 3) and torch.equal(d.mode(), lg.argmax(-1))


def test_agent_action_vocabulary(monkeypatch):
    """Stretch action list of the evaluation agent: order of the policy head (stretch_initialization_utils.py:145-166), long-name
    override like upstream."""
    from safevla_amd import agent
    a = agent.InferenceAgentVIDA.__new__(agent.InferenceAgentVIDA)
    monkeypatch.delenv("ACTION_DICT", raising=False)
    monkeypatch.delenv("LONG_ACTION_NAME", raising=False)
    names = a.get_action_list()
    assert len(names) == 20 and len(set(names)) == 20 and names[:5] == ["m", "r", "l", "b", "end"] and names[-1] == "d"
    monkeypatch.setenv("LONG_ACTION_NAME", "1")
    assert a.get_action_list()[:3] == ["move_ahead", "rotate_right", "rotate_left"]


def test_goal_tokenizer_sentencepiece_hook(tmp_path):
    """the real text path is a sentencepiece model (t5-small's spiece.model: not available offline).  The hook
    is exercised with a tiny unigram model trained here: ids come from sentencepiece, EOS (= 1, T5 convention) is appended, batches
    are padded with 0 and masked like HF's ``padding=True``."""
    spm = pytest.importorskip("sentencepiece")
    from safevla_amd.text import EOS_ID, PAD_ID, GoalTokenizer, bytes_to_str, str_to_bytes

    corpus = tmp_path / "corpus.txt"
    goals = ["find a mug", "navigate to the red chair and pick up the cup", "go to the kitchen", "pick up the apple on the table",
             "fetch a bowl from the living room", "locate a houseplant"]
    corpus.write_text("\n".join(goals * 20))
    prefix = str(tmp_path / "toy")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=prefix, vocab_size=48, model_type="unigram", pad_id=0, eos_id=1, unk_id=2,
                                   bos_id=-1, hard_vocab_limit=False, minloglevel=2)
    tok = GoalTokenizer(spiece_model=prefix + ".model")
    sp = spm.SentencePieceProcessor(model_file=prefix + ".model")
    for gtxt in goals[:3]:
        ids = tok.encode(bytes_to_str(str_to_bytes(gtxt, 1000).reshape(-1)))          # through the byte-string sensor format
        assert ids[-1] == EOS_ID and ids[:-1] == list(sp.encode(gtxt)) and all(i > PAD_ID for i in ids)
        assert sp.decode(ids[:-1]) == gtxt
    enc = tok(goals[:3])
    assert enc["input_ids"].shape == enc["attention_mask"].shape
    for row, m in zip(enc["input_ids"], enc["attention_mask"]):
        n = int(m.sum())
        assert (row[n:] == PAD_ID).all() and row[n - 1] == EOS_ID and (m[:n] == 1).all()


# ---- round 2 ---------------------------------------------------------------------------------------------------------------------
def test_safe_rl_step_result_and_box_containers():
    """Env-step container of the reference's Task.step (tasks/abstract_task.py:369-381) and the preprocessor observation_space stand-in."""
    from safevla_amd.api import Box, SafeRLStepResult

    r = SafeRLStepResult(observation={"x": 1}, reward=10.0, cost=2, done=True, info={"action": "end"})
    assert r._fields == ("observation", "reward", "cost", "done", "info")
    assert r.clone({"cost": 0}).cost == 0 and r.clone({"cost": 0}).reward == 10.0
    assert r.merge(SafeRLStepResult(None, None, 5, None, None)) == SafeRLStepResult({"x": 1}, 10.0, 5, True, {"action": "end"})
    assert Box(-float("inf"), float("inf"), (84, 384)).shape == (84, 384)


def test_mixed_task_assignment_follows_global_env_index():
    from safevla_amd.synth_env import MIXED_ORDER, env_tasks

    full = env_tasks("Mixed", 256)
    assert full[:6] == ["ObjectNav", "PickUp", "Fetch"] * 2 and {full.count(t) for t in MIXED_ORDER} <= {85, 86}
    # sharding the 256 envs of C5 over 8 ranks keeps "env e -> task e mod 3"
    stitched = []
    for rank in range(8):
        s, n = parallel.shard_envs(256, 8, rank)
        stitched += env_tasks("Mixed", n, env_offset=s)
    assert stitched == full
    assert env_tasks("Fetch", 3) == ["Fetch"] * 3


def test_lightning_checkpoint_accepts_path_dict_and_bare_state_dict(tmp_path):
    """checkpoint.load_pl_ckpt_allenact (training/offline/train_utils.py:6-68): ``model.`` prefix, actor.weight -> actor.linear.weight,
    image-encoder keys ignored -- for a path, a whole Lightning checkpoint and its bare state dict (build_agent's branch)."""
    from safevla_amd.checkpoint import load_pl_ckpt_allenact

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.actor = torch.nn.Module()
            self.actor.linear = torch.nn.Linear(4, 3)
            self.other = torch.nn.Linear(2, 2)

    sd = {"model.actor.weight": torch.full((3, 4), 0.5), "model.actor.bias": torch.full((3,), -1.0),
          "model.visual_encoder.image_encoder.model.x": torch.zeros(1), "model.unknown": torch.zeros(2)}
    path = str(tmp_path / "il.ckpt")
    torch.save({"state_dict": sd}, path)
    for arg in (path, {"state_dict": sd}, sd):
        m = M()
        loaded, missing, extra = load_pl_ckpt_allenact(m, arg)
        assert (m.actor.linear.weight == 0.5).all() and (m.actor.linear.bias == -1).all()
        assert sorted(loaded) == ["actor.linear.bias", "actor.linear.weight"] and sorted(missing) == ["other.bias", "other.weight"]
        assert extra == ["unknown"]


def test_bench_refuses_to_run_fewer_ranks_than_requested():
    """`python bench.py --gpus N` must never silently run one rank: on a box with fewer GPUs it exits non-zero."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and "refusing to run fewer ranks" in (r.stderr + r.stdout)
    env["WORLD_SIZE"] = "2"; env["RANK"] = "0"
    src = open(os.path.join(root, "bench.py")).read()
    assert "or world == 1" not in src


def test_reference_entry_point_shim_parses_fire_style_command_lines():
    import importlib.util
    import os

    from safevla_amd.train import build_parser, infer_task

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("entry", os.path.join(root, "training", "online", "dinov2_vits_tsfm_base.py"))
    entry = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(entry)
    with pytest.raises(SystemExit):
        entry.main(["--num_train_processes", "4"])              # no command
    with pytest.raises(SystemExit):
        entry.main(["test", "--checkpoint", "x"])              # simulator-bound evaluation runner: refused with a pointer to the agent
    a = build_parser().parse_args(["train", "--il_ckpt_path=il.ckpt", "--num_train_processes", "32", "--dataset_dir", "data/fifteen/FetchType",
                                   "--tag", "FetchType", "--cost_limit", "2.31964", "--wandb_project", "p", "--wandb_entity", "e",
                                   "--callbacks", "wandb_logging_callback", "--checkpoint", "c.pt"])
    assert (a.il_ckpt_path, a.num_train_processes, a.cost_limit, a.checkpoint) == ("il.ckpt", 32, 2.31964, "c.pt")
    assert infer_task(a.tag, a.dataset_dir) == "Fetch" and infer_task("ObjectNavType", "") == "ObjectNav" and infer_task("PickupType", "") == "PickUp"


def test_polynomial_erf_gelu_of_the_kernels_against_scipy():
    """safevla_amd/asmgen/gelu_poly.py: the degree-9 form every GELU epilogue evaluates (HIP: csrc/gemm.hip gelu_f through the generated header; assembly:
    svla_nt_as_k384_f2) against the exact erf-GELU of nn.GELU() (the frozen ViT's Mlp), over all magnitudes; exact tails; the generated C header carries
    the same fp32 constants."""
    import re
    import struct

    import numpy as np
    from scipy.special import erf

    from safevla_amd.asmgen import gelu_poly as G

    x = np.concatenate([np.linspace(-12, 12, 600001), np.array([-1e4, -50.0, 50.0, 1e4, 0.0, -0.0])]).astype(np.float32)
    y = G.gelu_ref_np(x)
    exact = x.astype(np.float64) * 0.5 * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    assert np.abs(y - exact).max() < 3e-5
    assert (y[x >= G.CLAMP] == x[x >= G.CLAMP]).all() and (y[x <= -G.CLAMP] == 0).all()
    hdr = G.c_header()
    cs = [float(v) for v in re.search(r"SVLA_GELU_COEFS \{(.*)\}", hdr).group(1).replace("f", "").split(",")]
    assert [struct.unpack("<I", struct.pack("<f", v))[0] for v in cs] == G.COEF_BITS


def test_il_preset_table_matches_the_reference_source():
    """il.EarlyFusionCnnTransformer.VERSIONS against the reference's own preset table, read from its source where the build container has it
    (early_fusion_tsfm_models.py:221-312): layers / width / heads of the fusion transformer and of the decoder, image and text encoder of every
    preset that the reference can construct."""
    import os
    import re
    path = "/root/reference/architecture/models/transformer_models/early_fusion_tsfm_models.py"
    if not os.path.exists(path):
        pytest.skip("the reference tree is only present in the build container")
    from safevla_amd.il import EarlyFusionCnnTransformer as M
    src = open(path).read()
    src = src[src.index("def build_model"):src.index("model = EarlyFusionCnnTransformer(model_cfg)")]
    feat = {"Dinov2Small": 384, "Dinov2Base": 768, "SigLIPBase": 768, "SigLIPLarge": 1024, "ClipResNet50": 2048}
    seen = {}
    for blk in re.split(r"\n\s+(?:if|elif) model_version == ", src)[1:]:
        names = re.findall(r'"([a-zA-Z0-9_]+)"', blk.split(":")[0])
        img = re.search(r'image_encoder = "(\w+)"', blk).group(1)
        txt = re.search(r'text_encoder = "([\w-]+)"', blk).group(1)
        fus = re.search(r"fusion_xformer = TransformerConfig\((\d+), (\d+), (\d+)\)", blk)
        dec = re.search(r"model_cfg.decoder = TransformerConfig\((\d+), (\d+), (\d+)\)", blk)
        for n in names:
            seen[n] = (img, txt, tuple(map(int, fus.groups())) if fus else None, tuple(map(int, dec.groups())))
    assert len(seen) == 19
    dead = {"small_3_nonTxEnc", "siglip_base_3_nonTxEnc", "siglip_base_384_3", "siglip_base_384_resize_3"}     # cannot be constructed in the reference (il.VERSIONS)
    assert set(seen) - dead == set(M.VERSIONS)
    for n in M.VERSIONS:
        img, txt, fus, dec = seen[n]
        nf, nd, dd, te, dm, nh, nhd = M.version_config(n)
        assert (nf, dm, nh) == fus and (nd, dm, nhd) == dec and dd == feat[img] and te == txt, n


def test_bench_flop_accounting_matches_the_survey():
    """bench.py's reference-schedule FLOPs per update (the denominator of every "fraction of peak" on the reference's schedule) == SURVEY.md section 8(d):
    C2 (4 096 rows, S = 181) 573 TFLOP, C3 (16 384 rows) 2.29 PFLOP per update of 3 towers x (fwd + 2 bwd) x 4 epochs; linear in rows (up to the per-goal text adapter) and in epochs."""
    import bench

    c2, c3 = bench.flops_per_update(4096, 181, 12, 32, 4), bench.flops_per_update(16384, 181, 12, 64, 4)
    assert abs(c2 / 1e12 - 573) < 1.0 and abs(c3 / 1e15 - 2.29) < 0.005
    assert abs(c3 / c2 - 4.0) < 1e-4 and abs(      # (the text adapter is counted per unique goal, not per row)
               bench.flops_per_update(4096, 181, 12, 32, 1) * 4 - c2) < 1e-6 * c2
    assert bench.flops_per_update(4096, 233, 64, 32, 4) > 1.25 * c2        # 64-token instructions: S = 233
