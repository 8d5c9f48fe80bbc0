"""In-tree build of the MI355X (gfx950) SpMM hot path.

Produces, next to this file:
  libmi_spmm.so                      the C-ABI library (include/mi_spmm.h): every
                                     csrc/*.hip compiled by hipcc for gfx950
  custom_mm.cpython-*.so             the pybind11 module `custom_mm` (csrc/custom_mm.cpp)
                                     that mirrors the reference's src/custom_mm.cpp surface

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only build
container; the built files travel to the GPU box with the tree.

    python matrix-multiplication_amd/build.py [--force] [--verbose]        (MI_BUILD_FORCE=1 = --force, also through
                                                                            __graft_entry__.build(): objects are otherwise
                                                                            reused when their sources' hashes match)
"""
from __future__ import annotations

import argparse
import hashlib
import os
import re
import shlex
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
INCLUDE = REPO / "include"
OBJ_DIR = PKG_DIR / "build"
LIB_PATH = PKG_DIR / "libmi_spmm.so"
ARCH = "gfx950"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
HIPCC = str(ROCM / "bin" / "hipcc")


def ext_path() -> Path:
    return PKG_DIR / ("custom_mm" + sysconfig.get_config_var("EXT_SUFFIX"))


def _run(cmd, verbose):
    if verbose:
        print("+", " ".join(shlex.quote(str(c)) for c in cmd), flush=True)
    proc = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    if proc.returncode != 0:
        sys.stderr.write(proc.stdout)
        sys.stderr.write(proc.stderr)
        raise RuntimeError("build step failed: " + " ".join(str(c) for c in cmd))
    if verbose and proc.stderr.strip():
        sys.stderr.write(proc.stderr)


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(str(x) for x in paths):
        h.update(p.encode())
        h.update(Path(p).read_bytes())
    return h.hexdigest()


def _stale(target: Path, stamp: Path, digest: str) -> bool:
    return not target.exists() or not stamp.exists() or stamp.read_text() != digest


def build_library(force=False, verbose=False) -> Path:
    """hipcc --offload-arch=gfx950 every csrc/*.hip → libmi_spmm.so."""
    OBJ_DIR.mkdir(exist_ok=True)
    sources = sorted(CSRC.glob("*.hip"))
    headers = sorted(CSRC.glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", f"-I{INCLUDE}", f"-I{CSRC}",
             "-Wall", "-Wno-unused-function"]
    flags += shlex.split(os.environ.get("MI_HIPCC_FLAGS", ""))  # developer A/B builds (-DMI_…); part of the stamp
    hdr_digest = _digest(headers, " ".join(flags))

    def compile_one(src: Path):
        obj = OBJ_DIR / (src.stem + ".o")
        stamp = OBJ_DIR / (src.stem + ".stamp")
        # a unit that includes another .hip (gemm_f32_nn.hip → gemm_f32.hip) is stale when that file changes
        included = [CSRC / m for m in re.findall(r'#include "([\w.]+\.hip)"', src.read_text()) if (CSRC / m).exists()]
        digest = _digest([src, *included], hdr_digest)
        if force or _stale(obj, stamp, digest):
            _run([HIPCC, *flags, "-c", src, "-o", obj], verbose)
            stamp.write_text(digest)
            return obj, True
        return obj, False

    with ThreadPoolExecutor(max_workers=min(max(2, (os.cpu_count() or 4) - 2), len(sources))) as pool:
        results = list(pool.map(compile_one, sources))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or not LIB_PATH.exists():
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB_PATH], verbose)
    return LIB_PATH


def build_extension(force=False, verbose=False) -> Path:
    """g++ csrc/custom_mm.cpp → custom_mm extension module linked to libmi_spmm.so and torch."""
    import torch  # noqa: F401  (needed for include / library paths)
    from torch.utils import cpp_extension as ce

    OBJ_DIR.mkdir(exist_ok=True)
    src = CSRC / "custom_mm.cpp"
    out = ext_path()
    stamp = OBJ_DIR / "custom_mm.stamp"
    torch_lib = Path(torch.__file__).resolve().parent / "lib"
    inc = [f"-I{p}" for p in ce.include_paths("cuda")] + [f"-I{INCLUDE}", f"-I{CSRC}", f"-I{sysconfig.get_paths()['include']}"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    flags = ["-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
             "-DTORCH_EXTENSION_NAME=custom_mm", "-DTORCH_API_INCLUDE_EXTENSION_H",
             f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-Wno-deprecated-declarations"]
    link = [f"-L{PKG_DIR}", "-lmi_spmm", f"-L{torch_lib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
            "-ltorch", "-ltorch_python", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{torch_lib}"]
    digest = _digest([src, *sorted(CSRC.glob("custom_mm_*.inc")), INCLUDE / "mi_spmm.h"], " ".join(flags + link) + torch.__version__)
    if force or _stale(out, stamp, digest):
        _run(["g++", *flags, *inc, src, "-o", out, *link], verbose)
        stamp.write_text(digest)
    return out


def build_all(force=False, verbose=False):
    # MI_BUILD_FORCE=1: recompile everything whatever the stamps say (a driver that wants the build exercised, not reused)
    force = force or os.environ.get("MI_BUILD_FORCE", "") == "1"
    lib = build_library(force, verbose)
    ext = build_extension(force, verbose)
    return lib, ext


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--lib-only", action="store_true")
    a = ap.parse_args()
    if a.lib_only:
        print(build_library(a.force, a.verbose))
    else:
        for p in build_all(a.force, a.verbose):
            print(p)
