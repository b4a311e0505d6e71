"""Build the gfx950 C-ABI shared library (libsvla_hip.so) in-tree with hipcc.  No torch, no JIT cache."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsvla_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA results land in arch VGPRs (gfx950's register file is unified) instead of AGPRs, which
# removes ~5 v_accvgpr_read/write moves per MFMA in the attention kernels where VALU code consumes the tiles directly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Wall", "-Wno-unused-function"]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src, *extra))


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        if force or _newer(s, o, hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC, *FLAGS, "-I", os.path.join(HERE, "..", "include"), "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, os.path.basename(s) + ".o") for s in srcs]
    if force or jobs or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
