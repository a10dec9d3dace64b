"""The C ABI library must load and export every function include/graphaligner_amd.h declares.
No compute calls are made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "graphaligner_amd.h")
LIB = os.path.join(ROOT, "graphaligner_amd", "libgraphaligner_amd.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ga_[a-z_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ["ga_graph_create", "ga_graph_add_node", "ga_graph_add_edge", "ga_graph_finalize", "ga_graph_upload", "ga_align_batch",
                 "ga_results_free", "ga_batch_prepare", "ga_batch_run", "ga_batch_collect"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import __graft_entry__ as entry
        entry.build_product()
    lib = ctypes.CDLL(LIB)
    for name in declared_functions():
        assert hasattr(lib, name), "libgraphaligner_amd.so does not export %s" % name


def test_no_cpu_fallback_without_device():
    """graph upload must fail loudly when no GPU is usable (this container has none)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from graphaligner_amd import binding
    with pytest.raises(RuntimeError):
        binding.Graph([(1, "ACGTACGTAC")], [])


def test_product_library_does_not_contain_the_oracle_or_the_emulation():
    data = open(LIB, "rb").read()
    assert b"gao_align" not in data and b"ga_emul_" not in data
