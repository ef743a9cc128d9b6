"""Build recipe for the C-ABI HIP library (``libcookietts_hip.so``), in-tree, gfx950 only.

``hipcc`` cross-compiles without a GPU, so this runs in the build container; the built
``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(PKG_DIR, "libcookietts_hip.so")
OBJ_DIR = os.path.join(PKG_DIR, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("CTTS_HIPCC_EXTRA", "").split()
# per-file additions.  tacotron_persistent.hip: the SLP vectoriser packs its scalar fp32 FMAs into v_pk_fma_f32, whose
# register-pair operands cost hundreds of moves and spills in a kernel that keeps ~100 weights resident per lane
EXTRA_FLAGS = {"tacotron_persistent.hip": ["-fno-slp-vectorize"]}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the cookietts HIP library cannot be built")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return hdrs


_INC = None


def _includes(path, seen=None):
    """Transitive ``#include "..."`` closure of one source (headers of csrc/ and include/): a header edit rebuilds only
    the objects that see it."""
    import re
    global _INC
    _INC = _INC or re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
    seen = set() if seen is None else seen
    try:
        text = open(path).read()
    except OSError:
        return seen
    for inc in _INC.findall(text):
        cand = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if not os.path.exists(cand):
            cand = os.path.join(INCLUDE, os.path.basename(inc))
        if os.path.exists(cand) and cand not in seen:
            seen.add(cand)
            _includes(cand, seen)
    return seen


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in sources() + _deps() + [os.path.abspath(__file__)])


def build(force=False, verbose=True):
    """Compile every .hip under csrc/ for gfx950 and link the shared library."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        newest = max([os.path.getmtime(src), os.path.getmtime(os.path.abspath(__file__))] +
                     [os.path.getmtime(h) for h in _includes(src)])
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest:
            continue
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-I", INCLUDE, "-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
