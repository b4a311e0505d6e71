"""Register / spill budget of the hot HIP kernels, checked at compile time (hipcc cross-compiles gfx950 without a GPU).

Round 6 found a 5.6 % regression of the whole update by an A/B against the previous round's tree: wrapping the attention forward kernels' bodies in same-named
__global__ functions (for the tower-grouped launches) made `attn_fwd_persist_kernel<12>` spill 109 VGPRs instead of 6 and run 2x slower, while every parity test stayed
green.  This test pins what the compiler does with the kernels that carry the update: VGPR count (= occupancy) and spill counts from -Rpass-analysis=kernel-resource-usage."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "safevla_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-c"]

# kernel-name fragment (mangled) -> (max VGPRs, max spilled VGPRs): the values of the round-6 build with a little slack; a change of schedule that needs more is a
# decision to take with an A/B in hand (tools/ab_attn.py, bench.py against the previous tree), not something to discover in a profile a round later
BUDGET = {
    "attn.hip": {
        "28attn_fwd_persist_kernel_bodyILi12E": (256, 12),             # the fusion attention forward of the update (2 workgroups per CU)
        "27attn_bwd_fused_exact_kernelILi12ELb1E": (170, 8),           # its backward with dropout: THREE workgroups per CU need <= 170
        "27attn_bwd_fused_exact_kernelILi12ELb0E": (170, 8),
        "20attn_fwd_kernel_bodyILi18ELb0ELi4ELb1E": (176, 0),          # the ViT at 224 x 224 (S = 257): two workgroups per CU
        "20attn_fwd_kernel_bodyILi28ELb0ELi8ELb1E": (216, 0),          # the ViT's S = 433 attention
    },
    "gemm.hip": {
        "19gemm_nt_bf16_kernel10GemmNtArgs": (192, 0),                 # 128-tile kernel (two workgroups per CU)
        "21gemm_nt8p_bf16_kernelILi0ELi1ELb1EE": (256, 0),             # linear2 / out_proj forward with residual + dropout
        "21gemm_nt8p_bf16_kernelILi1ELi0ELb1EE": (256, 0),
        "26gemm_nt8p_bf16_kernel_bodyILi0ELi0ELb0EE": (256, 0),        # grouped twins must match their single-launch kernels
        "24gemm_nt_bf16_kernel_body10GemmNtArgs": (192, 0),
    },
    "norm.hip": {
        "15norm_fwd_kernelItLi512EE": (64, 0),
        "15norm_bwd_kernelItLi512EE": (128, 4),
    },
}


def _resources(src):
    r = subprocess.run([HIPCC, *FLAGS, "-I", os.path.join(ROOT, "include"), os.path.join(CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True, cwd=CSRC)
    assert r.returncode == 0, r.stderr[-3000:]
    out, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark: [^ ]+ +(VGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_hot_kernels_keep_their_register_budget():
    with ThreadPoolExecutor(max_workers=len(BUDGET)) as ex:
        res = dict(zip(BUDGET, ex.map(_resources, BUDGET)))
    bad = []
    for src, kernels in BUDGET.items():
        for frag, (max_vgpr, max_spill) in kernels.items():
            hits = {k: v for k, v in res[src].items() if frag in k}
            assert hits, (src, frag, sorted(res[src])[:5])
            for k, v in hits.items():
                if v.get("VGPRs", 0) > max_vgpr or v.get("VGPRs Spill", 0) > max_spill:
                    bad.append((k[:90], v, (max_vgpr, max_spill)))
    assert not bad, bad
