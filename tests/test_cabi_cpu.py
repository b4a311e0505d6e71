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
