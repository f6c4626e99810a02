"""In-tree build of the native pieces (hipcc for gfx950, g++ for host-only C++).

``python -m flooder_amd.build`` or ``flooder_amd.build.build_all()``.  Outputs live next to the
package (``flooder_amd/libflooder_hip.so``, ``flooder_amd/libflooder_host.so``); they are
git-ignored but travel to the GPU box with the working tree.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
HIP_LIB = os.path.join(PKG_DIR, "libflooder_hip.so")
HOST_LIB = os.path.join(PKG_DIR, "libflooder_host.so")
# CPython-API helpers (loaded with ctypes.PyDLL): built against ONE interpreter's headers with non-limited-API macros,
# so the file name carries that interpreter's ABI tag - another CPython sharing the checkout builds its own
import sysconfig as _sysconfig

PY_LIB = os.path.join(PKG_DIR, f"libflooder_py.{_sysconfig.get_config_var('SOABI') or 'abi-unknown'}.so")


def _tmp(path: str) -> str:
    """Per-process temporary name next to ``path``: ranks of one fresh multi-rank start all build, and os.replace
    must never publish another process's half-written file."""
    return f"{path}.{os.getpid()}.tmp"

HIP_SOURCES = ["flood_kernels.hip", "flood_bvh.hip", "flood_cell.hip", "flood_finish.hip", "flood_fps.hip", "flood_fps2.hip", "flood_index.hip", "flood_f64.hip", "flood_sorted.hip", "flood_wit.hip", "flood_params.hip"]
HOST_SOURCES = ["persistence.cpp", "delaunay3d.cpp", "delaunay2d.cpp", "delaunay_nd.cpp", "cell_faces.cpp"]
HOST_HEADERS = ["exact_int.hpp", "host_parallel.hpp"]


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libflooder_hip.so)")


def build_hip(force: bool = False, verbose: bool = False, out: str = None) -> str:
    """``out``: write the library there instead of the product path (diagnostic builds with FLOODER_HIPCC_FLAGS;
    load them with FLOODER_HIP_LIB=<out>)."""
    if out:
        return _build_hip_to(os.path.abspath(out), True, verbose)
    return _build_hip_to(HIP_LIB, force, verbose)


def _build_hip_to(HIP_LIB: str, force: bool, verbose: bool) -> str:
    """One object per source (compiled in parallel, kept under csrc/_obj keyed by the flags), then one link."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor

    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    hdrs = [os.path.join(ROOT, "include", "flooder_hip.h"), os.path.join(CSRC, "flood_common.hpp"),
            os.path.join(CSRC, "flood_bvh.hpp"), os.path.join(CSRC, "flood_planes.hpp")]
    if not force and _newer(HIP_LIB, srcs + hdrs):
        return HIP_LIB
    extra = os.environ.get("FLOODER_HIPCC_FLAGS", "").split()
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-DFLOODER_BUILD", *extra]
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10]
    obj_dir = os.path.join(CSRC, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(obj_dir, f"{os.path.basename(src)}.{tag}.o")
        if force or not _newer(obj, [src] + hdrs):
            cmd = [hipcc, *flags, "-c", src, "-o", _tmp(obj)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
            os.replace(_tmp(obj), obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as pool:
        objs = list(pool.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-o", _tmp(HIP_LIB)] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(_tmp(HIP_LIB), HIP_LIB)
    return HIP_LIB


def build_host(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not srcs:
        return ""
    if not force and _newer(HOST_LIB, srcs + [os.path.join(CSRC, h) for h in HOST_HEADERS]):
        return HOST_LIB
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        raise RuntimeError("g++ not found")
    # (no -march: the library travels to other hosts; delaunay_nd.cpp carries AVX2 / AVX-512 clones of its one hot loop
    # and picks at run time)
    cmd = [cxx, "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-o", _tmp(HOST_LIB)] + srcs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(_tmp(HOST_LIB), HOST_LIB)
    return HOST_LIB


def build_py(force: bool = False, verbose: bool = False) -> str:
    """The dict hand-off (csrc/pyhandoff.c): plain C against this interpreter's headers; no libpython link - the
    symbols resolve in the running interpreter."""
    import sysconfig

    src = os.path.join(CSRC, "pyhandoff.c")
    if not force and _newer(PY_LIB, [src]):
        return PY_LIB
    cc = shutil.which("gcc") or shutil.which("cc")
    inc = sysconfig.get_paths()["include"]
    if cc is None or not os.path.exists(os.path.join(inc, "Python.h")):
        return ""   # (no compiler or no headers: simplex_tree.to_dict falls back to the Python loop)
    cmd = [cc, "-O2", "-shared", "-fPIC", f"-I{inc}", "-o", _tmp(PY_LIB), src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(_tmp(PY_LIB), PY_LIB)
    return PY_LIB


def build_all(force: bool = False, verbose: bool = False):
    build_py(force, verbose)
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    if "--out" in sys.argv:   # python -m flooder_amd.build --out /tmp/libflooder_hip_timers.so  (diagnostic variant)
        print(build_hip(verbose=True, out=sys.argv[sys.argv.index("--out") + 1]))
    else:
        print(build_all(force="--force" in sys.argv, verbose=True))
