"""Randomised differential test inside the suite: tools/fuzz_kernels.py (GEMM NT with every epilogue flavour, GEMM TN, attention forward + backward with the three mask
kinds, LayerNorm / RMSNorm) on a fixed seed -- row counts around every tile and panel boundary, every N / K the argument checks accept -- against plain fp32 / fp64 torch on the
same operands.  The fixed-shape kernel tests pin the model's shapes; this pins the dispatcher (assembly kernels + ragged-tail launches, mid-M launches, 256- and 128-tile kernels)
on shapes nobody listed.  Round 5 ran six other seeds x 1 500 cases on the MI355X without a mismatch (profiles/r05_fuzz_kernels.txt)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11])
def test_fuzz_kernels_fixed_seed(seed):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_kernels.py"), "--seed", str(seed), "--cases", "500"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "0 failure line(s) in total" in r.stdout
    # the campaign reached the assembly kernels, both tile kernels and every mask kind
    for key in ("svla_nt_as_", "svla_nt_os_", "gemm_nt8p_bf16_kernel", "gemm_nt_bf16_kernel", "svla_tn_os", "causal", "t5"):
        assert key in r.stdout, (key, r.stdout[-3000:])


def test_fuzz_engine_fixed_seed():
    """tools/fuzz_engine.py: random (T, B, L, task, env-chunk, minibatch count) rollouts through the engine on the bf16 product path vs the fp32 verification mode, then a full
    update each; the first two configurations are ONE-step rollouts (round 5: a T = 1 update batch used to take the decoder's KV-cached acting branch and die in the backward)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_engine.py"), "--seed", "3", "--cases", "12"], cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "0 failing configuration(s) of 12" in r.stdout and r.stdout.count("ok   T=1 ") >= 2, r.stdout[-3000:]


def test_fuzz_acting_across_the_cache_window():
    """tools/fuzz_acting.py: models with a 5 / 8 / 17-slot KV window stepped through up to three windows of single-step forwards (episodes shorter than the window and outliving
    it): recorded launch plans == eager issue step for step, bf16 path vs fp32 mode on the ladder -- the counter wrap, which a dozen steps inside the 500-slot window never reach."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_acting.py"), "--seed", "2", "--cases", "6"], cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "0 failing rollout(s) of 6" in r.stdout, r.stdout[-3000:]


def test_fuzz_vit_batch_invariance():
    """tools/fuzz_vit.py: a frame's features do not depend on its batch -- 1 ... 130 frames through the frozen DINOv2 ViT-S/14 (two frame geometries) and the SigLIP ViT-B/16 trunk
    (padded token rows, tile / panel / mid-M kernels by row count) against the same frames alone."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_vit.py"), "--seed", "1"], cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 failing batch size(s)" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
