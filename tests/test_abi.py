"""CPU: libprf.so loads and exports every entry point include/prf.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

from conftest import PKG, ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "prf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prf_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for name in ("prf_open", "prf_close", "prf_scan", "prf_scan_genome", "prf_genome_load", "prf_free_hits",
                 "prf_last_error", "prf_abi_version", "prf_measure_hbm_read"):
        assert name in syms


def test_library_exports_every_declared_symbol():
    path = os.path.join(PKG, "libprf.so")
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), f"libprf.so does not export {name}"
    lib.prf_abi_version.restype = ctypes.c_int
    assert lib.prf_abi_version() == 4


def test_binding_lists_the_same_symbols():
    import prf_native
    assert sorted(prf_native.EXPORTS) == declared_symbols()


def test_no_cpu_fallback_without_a_device():
    """On a box without a GPU the product path must fail loudly, not fall back."""
    import prf_native
    lib = prf_native.load_library()
    if lib.prf_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(prf_native.PrfError):
        prf_native.Context(0)
    import argparse
    import perfect_repeat_finder as prf
    prf_native._default_ctx.clear()
    with pytest.raises(prf_native.PrfError):
        prf.detect_repeats("ACACACACACAC", argparse.Namespace(min_motif_size=1, max_motif_size=6, min_repeats=3, min_span=9))
