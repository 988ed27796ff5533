"""In-tree build of libpgo.so (hipcc, gfx950 only).  No JIT cache: the .so lives next to this
file so that it travels with the repository snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libpgo.so")
SOURCES = ["host_graph.cpp", "structure.cpp", "comm.cpp", "solver_create.hip", "solver_lm.hip", "solver_pcg.hip", "solver_direct.hip",
           "solver_batch.hip", "solver_abi.hip", "solver_launch.hip"]
HEADERS = ["pgo_internal.h", "comm.h", "kernels.hip.h", "solo.hip.h", "direct.hip.h", "coarse.hip.h", "solver_handle.hip.h",
           os.path.join(ROOT, "include", "pgo.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libpgo.so cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, defines=(), out: str = LIB) -> str:
    """`defines` / `out`: experiment builds (scripts/exp_*.sh) next to the product library, selected with PGO_LIB"""
    if not force and not needs_build() and out == LIB:
        return LIB
    # one object per translation unit, compiled side by side (the device code of a unit = the kernels it launches), then linked
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-function",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + ["-D" + d for d in defines]
    with tempfile.TemporaryDirectory(prefix="pgo_build_") as tmp:
        def compile_one(src):
            obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
            cmd = [_hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), max(1, (os.cpu_count() or 2) // 2))) as pool:
            objs = list(pool.map(compile_one, SOURCES))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out + ".tmp", "-lrccl", "-pthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    os.replace(out + ".tmp", out)
    return out


def build_cli(force: bool = False) -> str:
    """The C++ host mirror of the reference's `main` (DATASET NUM_OUTLIER_LOOPS METHOD)."""
    out = os.path.join(HERE, "host", "main")
    src = os.path.join(HERE, "host", "main.cpp")
    if not os.path.exists(src):
        return ""
    hdrs = [os.path.join(HERE, "host", h) for h in os.listdir(os.path.join(HERE, "host")) if h.endswith(".h")]
    if (not force and os.path.exists(out) and
            all(os.path.getmtime(out) >= os.path.getmtime(p) for p in [src, LIB] + hdrs)):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "host"), src,
           "-o", out, "-L" + HERE, "-lpgo", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE]
    subprocess.check_call(cmd)
    return out
