"""ctypes binding of the C-ABI shared library (include/svla.h is the single source of truth).

The product path has NO fallback: if ``libsvla_hip.so`` is missing or an entry point returns non-zero, this raises.
"""
import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsvla_hip.so")
HEADER = os.path.join(HERE, "..", "include", "svla.h")

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double}


def parse_header(path: str = HEADER) -> Dict[str, List[Tuple[str, object]]]:
    """{symbol: [(arg_name, ctype), ...]} for every ``int svla_*(...)`` declaration."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\bint\s+(svla_\w+)\s*\((.*?)\)\s*;", src, flags=re.S):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            name = re.findall(r"(\w+)$", a)[0]
            ty = a[: -len(name)].strip()
            if "*" in ty:
                ct = ctypes.c_void_p
            else:
                base = ty.replace("const", "").strip()
                ct = _CT[base]
            args.append((name, ct))
        out[m.group(1)] = args
    return out


def abi_signature(path: str = HEADER) -> str:
    """ordered 'name(type,type,...)' list of the entry points svla_replay_calls dispatches by integer id (declaration order of the header, the
    replay entry itself excluded).  build.py bakes its hash into the generated dispatcher (svla_replay_abi_stamp); _Lib compares."""
    src = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    out = []
    for m in re.finditer(r"\bint\s+(svla_\w+)\s*\((.*?)\)\s*;", src, flags=re.S):
        if m.group(1).startswith("svla_replay_calls"):
            continue
        tys = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            name = re.findall(r"(\w+)$", a)[0]
            tys.append(a[: -len(name)].strip().replace(" ", ""))
        out.append(f"{m.group(1)}({','.join(tys)})")
    return ";".join(out)


def abi_stamp(path: str = HEADER) -> int:
    h = 0xcbf29ce484222325
    for b in abi_signature(path).encode():
        h = ((h ^ b) * 0x100000001b3) & 0xffffffffffffffff
    return h


def torch_cuda_available() -> bool:
    import torch

    return torch.cuda.is_available()


class SvlaError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise SvlaError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU / eager fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `python safevla_amd/build.py`."
            )
        # torch first: the PyTorch-ROCm wheel carries its own HIP runtime; loading this library before it would bind a second runtime
        # instance (launches on torch's streams then fail with hipErrorNoDevice) -- seen when build() and smoke() share one process
        import torch  # noqa: F401

        self.cdll = ctypes.CDLL(LIB_PATH)
        self.decls = parse_header()
        # ids of the entry points in svla_replay_calls: declaration order of the header, the replay entry itself excluded (build.py generates
        # the dispatcher from the same header in the same order)
        self.fn_ids = {n: i for i, n in enumerate(k for k in self.decls if not k.startswith("svla_replay_calls"))}
        for name, args in self.decls.items():
            fn = getattr(self.cdll, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = ctypes.c_int
            fn.argtypes = [ct for _, ct in args]
        # svla_replay_calls dispatches by integer id = position of the declaration in the header: a library built from another revision of the
        # header would call the wrong kernel with reinterpreted arguments.  The generated dispatcher carries the hash of the ordered
        # signature list it was built from.
        stamp = ctypes.c_ulonglong(0)
        self.cdll.svla_replay_abi_stamp(ctypes.byref(stamp))
        self._gpu_init_done = False
        if stamp.value != abi_stamp():
            raise SvlaError(f"{LIB_PATH} was built from a different include/svla.h (entry-point table stamp {stamp.value:#x}, header {abi_stamp():#x}): "
                            "rebuild with `python safevla_amd/build.py`")

    recorder = None      # optional list: every call is appended as (bound C function, args) -- launch-replay experiments (tools/replay_probe.py)

    def call(self, name: str, *args):
        if not self._gpu_init_done:
            # one-time device-side initialisation of the GEMM dispatchers (CU count, zero bias, the assembly code object) before the first real launch -- not
            # lazily inside a HIP-graph capture or a replay thread.  Deferred to the first call so that CPU-only processes can still load the library.
            self._gpu_init_done = True
            if torch_cuda_available():
                rc = self.cdll.svla_gemm_force_small_tile(0)
                if rc != 0:      # (hipMalloc of the dispatchers' scratch or the load of the assembly code object failed: say so here, not as an opaque status of a later GEMM)
                    self._gpu_init_done = False
                    raise SvlaError(f"one-time GPU initialisation of libsvla_hip.so failed with status {rc} (svla_gemm_force_small_tile(0): CU count, zero-bias scratch, "
                                    "assembly code object)")
        fn = getattr(self.cdll, name)
        if self.recorder is not None:
            self.recorder.append((fn, args))
        rc = fn(*args)
        if rc != 0:
            raise SvlaError(f"{name} failed with status {rc}" + (" (invalid argument)" if rc == -1 else " (hipError_t)"))


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
