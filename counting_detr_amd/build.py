"""Build libcdetr_hip.so (hipcc, gfx950) in-tree: counting_detr_amd/lib/libcdetr_hip.so.

The .so travels to the GPU box with the repo snapshot (git-ignored, not gpurun-ignored).  hipcc cross-compiles
without a GPU.  Usage: python -m counting_detr_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcdetr_hip.so")
SOURCES = ["api.hip", "igemm.hip", "igemm_dl.hip", "rcda.hip", "mha.hip", "matcher.hip", "criterion.hip", "elementwise.hip", "glue.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def _deps():
    inc = os.path.join(os.path.dirname(HERE), "include", "cdetr_hip.h")
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))] + \
        [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "rows.h"), os.path.join(CSRC, "dl_common.h"), inc]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build_lib(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    headers = [d for d in _deps() if d.endswith(".h")]
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in [src] + headers):
            continue                                # this object is newer than its source and every header: keep it
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
