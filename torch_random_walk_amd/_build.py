"""In-tree build of the two native pieces (no JIT cache, nothing under ~/.cache).

* ``libtrw_hip.so``      hipcc, --offload-arch=gfx950: the kernels + the C ABI of include/trw.h
* ``torch_rw_native*.so`` g++: the pybind/torch host extension that calls that ABI

Both land next to this file so that a snapshot of the repository carries them to
a GPU box.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

HIP_LIB = os.path.join(HERE, "libtrw_hip.so")
EXT_LIB = os.path.join(HERE, "torch_rw_native" + sysconfig.get_config_var("EXT_SUFFIX"))

HIP_SOURCES = [os.path.join(CSRC, "trw_hip.hip"), os.path.join(CSRC, "trw_cpu.cpp")]
HIP_DEPS = HIP_SOURCES + [os.path.join(CSRC, f) for f in ("glibc_jump.hpp", "trw_kernels_wide.hpp",
                                                           "trw_kernels_narrow.hpp", "trw_kernels_marks.hpp", "trw_kernels_triples.hpp", "trw_kernels_edgelist.hpp", "trw_kernels_windows.hpp",
                                                           "trw_kernels_wire.hpp")] + [os.path.join(INCLUDE, "trw.h")]
EXT_SOURCES = [os.path.join(CSRC, "torch_rw_native.cpp")]
EXT_DEPS = EXT_SOURCES + [os.path.join(INCLUDE, "trw.h")]


def _digest(deps, extra=""):
    import hashlib
    h = hashlib.sha256(extra.encode())
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target, deps, extra=""):
    """Fresh <=> the stamp written next to `target` holds the digest of its sources' CONTENTS (mtimes do
    not survive a snapshot to another box; the stamp travels with the binary)."""
    stamp = target + ".stamp"
    if not (os.path.exists(target) and os.path.exists(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != _digest(deps, extra)


def _write_stamp(target, deps, extra=""):
    with open(target + ".stamp", "w") as f:
        f.write(_digest(deps, extra))


def _run(cmd):
    print("[torch_random_walk_amd build]", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.check_call(cmd)


def build_hip(force=False, extra_flags=()):
    if not force and not _stale(HIP_LIB, HIP_DEPS, " ".join(extra_flags)):
        return HIP_LIB
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    tmp = HIP_LIB + ".tmp"
    _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
          "-Wno-unused-function", *extra_flags, "-o", tmp, *HIP_SOURCES])
    os.replace(tmp, HIP_LIB)
    _write_stamp(HIP_LIB, HIP_DEPS, " ".join(extra_flags))
    return HIP_LIB


def build_ext(force=False):
    if not force and not _stale(EXT_LIB, EXT_DEPS):
        return EXT_LIB
    import torch
    from torch.utils import cpp_extension
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = []
    for p in cpp_extension.include_paths() + [sysconfig.get_paths()["include"], os.path.join(ROCM, "include")]:
        inc += ["-isystem", p]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    tmp = EXT_LIB + ".tmp"
    _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
          "-DTORCH_EXTENSION_NAME=torch_rw_native", "-DTORCH_API_INCLUDE_EXTENSION_H",
          "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi,
          *inc, "-o", tmp, *EXT_SOURCES,
          "-L" + HERE, "-l:libtrw_hip.so", "-L" + torch_lib, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu",
          "-ltorch_hip", "-ltorch_python",
          "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + torch_lib])
    os.replace(tmp, EXT_LIB)
    _write_stamp(EXT_LIB, EXT_DEPS)
    return EXT_LIB


def build_all(force=False):
    build_hip(force)
    build_ext(force)
    return HIP_LIB, EXT_LIB


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print(HIP_LIB)
    print(EXT_LIB)
