"""CPU tests (-m "not gpu") of the drop-in boundary: the C-ABI library loads, exports every
symbol include/gpak.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from gp_ss_ak_amd import _lib, gpak

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="gpak.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpak_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_list_the_same_symbols():
    assert header_functions() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions() + header_functions("gpak_dev.h"):
        assert hasattr(lib, name), name
    assert len(header_functions("gpak_dev.h")) == 30       # the device-pointer level API of the multi-GPU path
    from gp_ss_ak_amd import dist
    dist_fns = header_functions("gpak_dist.h")             # the C++ multi-GPU schedule
    assert sorted(dist.DIST_SYMBOLS) == dist_fns
    for name in dist_fns:
        assert hasattr(lib, name), name


def test_header_is_plain_c():
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "gpak.h"\n#include "gpak_dev.h"\n#include "gpak_dist.h"\n'
                             'int main(void){gpak_phase_times t; gpak_dist_stats s; (void)t; (void)s; return GPAK_OK;}\n')
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                               "-c", src, "-o", os.path.join(d, "t.o")])


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gpak.GpakError) as ei:
        gpak.Gpak(0)
    assert ei.value.status == gpak.EHIP
    assert "no CPU fallback" in str(ei.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gp_ss_ak_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".c")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f in ("synth.py",), os.path.join(dirpath, f)
