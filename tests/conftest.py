import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLD):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle runs next to the HIP path in most tests.  torch's CPU kernels get SLOWER beyond ~16 threads on the GPU boxes'
    # 256-thread hosts (bench.py's cpu_baseline sweep: 16 threads 0.4 s / step, 128 threads 6 s), so cap them for the whole session.
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is present, e.g. a plain ``pytest tests``."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# Measured errors that tests want in the driver's log even when they pass (pytest -q prints no captured stdout for passing
# tests): tests call ``conftest.report(...)``, the terminal summary prints the table.
_MEASURED = []


def report(what, measured, bound, note=""):
    _MEASURED.append((what, float(measured), float(bound), note))


def pytest_terminal_summary(terminalreporter):
    if not _MEASURED:
        return
    terminalreporter.section("measured errors (HIP path vs ground truth) | bound")
    for what, m, b, note in _MEASURED:
        terminalreporter.write_line("%-78s %.3e | %.3e %s" % (what, m, b, note))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLD, name + ".npz")))
        return cache[name]

    return load


def assert_close(actual, expected, rtol=1e-4, atol=1e-6, what=""):
    """rtol is the north-star 1e-4 relative bound; atol guards values that are ~0."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, "%s: shape %s vs %s" % (what, a.shape, e.shape)
    assert np.array_equal(np.isnan(a), np.isnan(e)), "%s: NaN pattern differs (%d vs %d NaNs)" % (what, np.isnan(a).sum(), np.isnan(e).sum())
    err = np.nan_to_num(np.abs(a - e), nan=0.0)
    tol = atol + rtol * np.nan_to_num(np.abs(e), nan=0.0)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: %d/%d out of tolerance; worst at %s: got %.9g want %.9g (|d|=%.3g, max|e|=%.3g)"
                             % (what, bad.sum(), bad.size, i, a[i], e[i], err[i], np.abs(e).max()))


def check_grad_compact(gold, key, actual, rtol=1e-4, atol=1e-6):
    """Counterpart of make_golden.put_grad."""
    a = np.asarray(actual)
    if key in gold:
        assert_close(a, gold[key], rtol, atol, key)
        return
    scale = float(gold[key + "@l2"])
    assert_close(a.reshape(-1)[::97], gold[key + "@s97"], rtol, atol + 1e-6 * scale, key + "@s97")
    assert abs(np.sqrt((a.astype(np.float64) ** 2).sum()) - scale) <= 1e-4 * scale + 1e-7, key + "@l2"
    want_sum = float(gold[key + "@sum"])
    assert abs(a.astype(np.float64).sum() - want_sum) <= 1e-4 * (scale + abs(want_sum)) + 1e-6, key + "@sum"


def assert_mostly_close(actual, expected, rtol, atol, what="", max_bad_frac=1e-2, agg_rtol=1e-3):
    """Per-element tolerance for all but a tiny fraction of elements, plus an aggregate (L1) bound.

    Used for gradient maps of the photometric loss: a pixel whose argmin / clamp / |.| branch sits
    within float rounding of a tie legitimately takes the other branch on a different summation
    order, which changes the gradient at that pixel (and its 3x3 neighbours) by O(1) relative."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, "%s: shape %s vs %s" % (what, a.shape, e.shape)
    err = np.abs(a - e)
    bad = err > atol + rtol * np.abs(e)
    frac = bad.mean()
    agg = err.sum() / max(np.abs(e).sum(), 1e-30)
    if os.environ.get("FD_TEST_VERBOSE"):
        print("[mostly_close] %s: %.4f%% out of tolerance, aggregate rel-L1 %.3g" % (what, 100 * frac, agg))
    if frac > max_bad_frac or agg > agg_rtol:
        i = np.unravel_index(np.argmax(err), err.shape)
        raise AssertionError("%s: %.4f%% of elements out of tolerance (allowed %.4f%%), aggregate rel-L1 %.3g "
                             "(allowed %.3g); worst at %s: got %.9g want %.9g"
                             % (what, 100 * frac, 100 * max_bad_frac, agg, agg_rtol, i, a[i], e[i]))


class _FdTune:
    """Handed to tests by the ``fdtune`` fixture: library thresholds (fd_set_tuning) and host-side issue switches for the duration of
    one test.  Replaces the FD_* environment variables of rounds 1-3 - the library no longer reads the environment."""

    def __init__(self):
        from fusiondepth_amd import tuning
        self._tuning = tuning
        self._lib0 = tuning.get_lib()
        self._host0 = {k: getattr(tuning.host, k) for k in dir(tuning.host) if not k.startswith("_")}

    def lib(self, **fields):
        self._tuning.set_lib(**fields)

    def host(self, **fields):
        for k, v in fields.items():
            if k not in self._host0:
                raise KeyError("tuning.host has no switch %r" % k)
            setattr(self._tuning.host, k, v)

    def restore(self):
        self._tuning.set_lib(**self._lib0)
        for k, v in self._host0.items():
            setattr(self._tuning.host, k, v)


@pytest.fixture
def fdtune():
    t = _FdTune()
    yield t
    t.restore()
