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


def test_library_behaviour_does_not_depend_on_the_environment(lib_path, monkeypatch):
    """kernel routing - and with it every fd_*_wt_floats / fd_*_ws_floats answer - is a function of the
    arguments and of fd_tuning alone.  The library does not import getenv at all; flipping the variables rounds 1-3 read changes
    nothing, fd_set_tuning does, and restoring the struct restores the answers."""
    import subprocess
    from fusiondepth_amd import _lib, tuning
    und = subprocess.run(["nm", "-D", "--undefined-only", lib_path], stdout=subprocess.PIPE, text=True).stdout
    assert "getenv" not in und
    srcs = os.path.join(ROOT, "fusiondepth_amd", "csrc")
    for f in os.listdir(srcs):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(srcs, f)).read(), f
    d = _lib.ConvDesc(12, 256, 12, 40, 256, 3, 3, 1, 1, 0, 0, 0)
    names = ("fd_conv2d_fwd_ws_floats", "fd_conv2d_fwd_wt_floats", "fd_conv2d_bwd_data_ws_floats", "fd_conv2d_bwd_data_wt_floats",
             "fd_conv2d_bwd_weight_ws_floats")
    sizes = lambda: tuple(_lib.query(n, ctypes.byref(d)) for n in names)
    base = sizes()
    for var in ("FD_WINO", "FD_WINO_FWD_2D", "FD_WINO_WGRAD_2D", "FD_WINO_FWD", "FD_WINO_WGRAD"):
        monkeypatch.setenv(var, "0")
    monkeypatch.setenv("FD_WINO_TARGET", "4096")
    assert sizes() == base
    gen = tuning.generation()
    with tuning.override(wino_fwd_2d_min=0, wino_wgrad_2d=0):
        assert tuning.generation() > gen
        changed = sizes()
        assert changed != base
    assert sizes() == base
    t = tuning.lib_defaults()
    assert t.size == ctypes.sizeof(tuning.Tuning) and tuning.get_lib() == {n: getattr(t, n) for n in tuning.LIB_FIELDS}
    with pytest.raises(KeyError):
        tuning.set_lib(no_such_field=1)
    bad = tuning.Tuning()
    _lib.load().fd_get_tuning(ctypes.byref(bad))
    bad.wino_target = 0
    assert _lib.load().fd_set_tuning(ctypes.byref(bad)) != 0 and "targets" in _lib.last_error()
    assert sizes() == base


def integration_stub():
    """The ctypes stub INTEGRATION.md shows a reference maintainer (section 2, first python block)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. C-ABI level"):]
    return re.search(r"```python\n(.*?)```", sec, flags=re.S).group(1)


def test_integration_stub_executes(lib_path, monkeypatch):
    """the stub asserted ``fd_abi_version() == 1`` against a header that says 2 - and compared an ``int``
    with ``b"gfx950"``.  It is executed here (the definitions and its two asserts; the GPU suite calls the functions it defines,
    tests/test_gpu_losspath.py::test_integration_stub_functions), so it cannot rot again."""
    from fusiondepth_amd import _lib
    monkeypatch.chdir(ROOT)
    code = integration_stub()
    assert "fd_abi_version() == %d" % _lib.ABI_VERSION in code
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    lib = ns["lib"]
    for name in set(re.findall(r"lib\.(fd_[a-z0-9_]+)", code)):
        assert hasattr(lib, name), name
        fn = getattr(lib, name)
        if fn.argtypes is not None and name in _lib.SIGNATURES:
            assert len(fn.argtypes) == len(_lib.SIGNATURES[name][0]), (name, len(fn.argtypes), _lib.SIGNATURES[name][0])
    assert callable(ns["ssim"]) and callable(ns["get_4beam_2channel"]) and callable(ns["get_4beam"])


def test_tuning_mirror_matches_the_header():
    """fusiondepth_amd.tuning.Tuning mirrors `fd_tuning` field for field, in order (the struct is append-only: a field inserted in the middle
    on one side only would silently shift every later threshold), and the library reports the mirror's size."""
    import ctypes
    import re
    from fusiondepth_amd import _lib, tuning
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "fdhip.h")).read()
    body = text[text.index("typedef struct fd_tuning {"):text.index("} fd_tuning;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"\bint\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [n for n, _ in tuning.Tuning._fields_], (names, [n for n, _ in tuning.Tuning._fields_])
    t = tuning.lib_defaults()
    assert t.size == ctypes.sizeof(tuning.Tuning) == 4 * len(names)
    assert set(tuning.LIB_FIELDS) == set(names) - {"size"}
    for var, (name, _) in tuning._ENV_LIB.items():
        assert name in names, (var, name)


def test_replay_dispatch_table_is_generated_from_the_signature_list():
    """csrc/replay_table.inc (the typed dispatch of fd_replay) is what gen_replay.py derives from _lib.SIGNATURES: a stale committed
    copy - an entry point added without regenerating - would dispatch recorded calls to the wrong function index"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_replay", os.path.join(root, "fusiondepth_amd", "gen_replay.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    from fusiondepth_amd import _lib
    text, names = gen.generate(_lib.SIGNATURES)
    with open(os.path.join(root, "fusiondepth_amd", "csrc", "replay_table.inc")) as fh:
        assert fh.read() == text
    lib = _lib.load()
    assert lib.fd_replay_function_count() == len(names)
    for i, n in enumerate(names):
        assert lib.fd_replay_function_name(i).decode() == n and lib.fd_replay_function_signature(i).decode() == _lib.SIGNATURES[n][0]
    assert "fd_conv2d_fwd" in names and "fd_bn_train_fwd" in names and "fd_set_tuning" not in names
