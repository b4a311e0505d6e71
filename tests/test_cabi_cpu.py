"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/svla.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from safevla_amd import build

    return build.build()


def test_library_exports_every_declared_symbol(built_lib):
    from safevla_amd._lib import parse_header

    decls = parse_header(os.path.join(ROOT, "include", "svla.h"))
    assert len(decls) >= 24
    cdll = ctypes.CDLL(built_lib)
    for name in decls:
        assert hasattr(cdll, name), f"{name} declared in include/svla.h but not exported"
    # and nothing with C linkage is exported that the header does not declare
    import subprocess

    syms = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (svla_\w+)", syms))
    assert exported == set(decls), exported ^ set(decls)


def test_header_cites_reference_for_every_entry_point():
    src = open(os.path.join(ROOT, "include", "svla.h")).read()
    # every declaration sits in a section whose comments cite the reference code it replaces (file:line) or mark
    # the un-vendored third-party origin [3P]
    for m in re.finditer(r"\bint (svla_\w+)\(", src):
        sec = src[src.rfind("/* ----", 0, m.start()) : m.start()]
        assert re.search(r"\.py:\d+|\[3P", sec), m.group(1)


def test_binding_refuses_cpu_tensors(built_lib):
    import torch

    from safevla_amd import ops

    with pytest.raises((ValueError, TypeError)):
        ops.gae_scan(*[torch.zeros(3, 2) for _ in range(4)], torch.zeros(4, 2), torch.zeros(2), torch.zeros(2))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "safevla_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M), f


def test_launch_group_capture_protocol_without_a_gpu(built_lib):
    """svla_group_begin / _member / _end / _stats (tower-grouped launches, csrc/launch.h) are host logic: the protocol -- one capture per thread, members in range,
    an empty capture issues nothing -- holds without a GPU; so do the argument checks of svla_det_set_grid and svla_acting_stage (refused before any launch)."""
    cdll = ctypes.CDLL(built_lib)
    assert cdll.svla_group_end(None) != 0                                 # nothing open
    assert cdll.svla_group_begin(0) != 0 and cdll.svla_group_begin(4) != 0
    assert cdll.svla_group_begin(3) == 0
    assert cdll.svla_group_begin(3) != 0                                  # captures do not nest
    assert cdll.svla_group_member(3) != 0 and cdll.svla_group_member(-1) != 0 and cdll.svla_group_member(2) == 0
    assert cdll.svla_group_end(None) == 0                                 # empty capture
    g, s = ctypes.c_long(-1), ctypes.c_long(-1)
    assert cdll.svla_group_stats(ctypes.byref(g), ctypes.byref(s)) == 0 and (g.value, s.value) == (0, 0)
    assert cdll.svla_det_set_grid(35) != 0 and cdll.svla_det_set_grid(53) != 0 and cdll.svla_det_set_grid(44) == 0 and cdll.svla_det_set_grid(52) == 0
    cdll.svla_acting_stage.restype = ctypes.c_int
    null = ctypes.c_void_p(0)
    args = [null, null, ctypes.c_long(0)] + [null] * 14 + [ctypes.c_int(0)] * 4 + [null] * 3 + [ctypes.c_int(0), null]
    assert cdll.svla_acting_stage(*args) != 0                             # B = 0 / null buffers: invalid, nothing launched
    from safevla_amd import ops
    assert [ops.det_grid_bits(n) for n in (64, 4096, 16384, 10 ** 6)] == [44, 50, 52, 52]
