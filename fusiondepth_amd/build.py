"""Build ``libfdhip.so`` (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting ``.so`` is
git-ignored but travels to the GPU box with the source snapshot.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ_DIR = os.path.join(CSRC, "_obj")
LIB_PATH = os.path.join(HERE, "libfdhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-Wno-pass-failed",
         "-I", INCLUDE, "-I", CSRC]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _file_flags(src):
    """Per-file compiler flags: a leading ``// FD_HIPCC_FLAGS: ...`` line of the source."""
    with open(src) as fh:
        first = fh.readline()
    return first.split(":", 1)[1].split() if first.startswith("// FD_HIPCC_FLAGS:") else []


def _compile(src, headers, force):
    obj = os.path.join(OBJ_DIR, os.path.basename(src).replace(".hip", ".o"))
    if force or _newer(obj, [src] + headers):
        cmd = [HIPCC] + FLAGS + _file_flags(src) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    sys.path.insert(0, HERE)
    try:
        import gen_replay                      # csrc/replay_table.inc from _lib.SIGNATURES (rewritten only when it changes)
        gen_replay.write()
    finally:
        sys.path.remove(HERE)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc"))) + sorted(glob.glob(os.path.join(INCLUDE, "*.h")))
    if not srcs:
        raise RuntimeError("no HIP sources under %s" % CSRC)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, headers, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = [os.path.basename(o) for o, changed in results if changed]
    if rebuilt or _newer(LIB_PATH, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-o", LIB_PATH] + objs, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
    if verbose:
        print("libfdhip.so: %s (%d sources, recompiled: %s)" % (LIB_PATH, len(srcs), ", ".join(rebuilt) or "none"))
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
