"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
that include/fdhip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "fdhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib_path():
    from fusiondepth_amd import build
    return build.build(verbose=False)


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in fdhip.h but not exported: %s" % missing


def test_python_binding_covers_the_header(lib_path):
    from fusiondepth_amd import _lib
    names = set(declared_functions())
    assert names == set(_lib.SIGNATURES), (sorted(names - set(_lib.SIGNATURES)), sorted(set(_lib.SIGNATURES) - names))
    lib = _lib.load()
    assert lib.fd_abi_version() == _lib.ABI_VERSION
    assert lib.fd_supported_arch() == b"gfx950"
    assert _lib.query("fd_photo_ws_floats", 6, 192, 640) == 6 * 12 * 10 * 4


def test_bad_arguments_return_an_error_not_a_crash(lib_path):
    from fusiondepth_amd import _lib
    with pytest.raises(RuntimeError, match="fd_scatter_2channel"):
        _lib.call("fd_scatter_2channel", None, None, 1, 192, 640, 76, 190, 2, 638, 2, None)
    assert "bad args" in _lib.last_error()


def test_ops_fail_loudly_without_gpu_tensors(lib_path):
    import torch
    from fusiondepth_amd import functional as FD
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="GPU"):
        FD.ssim(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="GPU"):
        FD.scatter_2channel(torch.zeros(1, 1, 192, 640))
