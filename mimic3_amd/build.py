"""Build the native library.

``build_hip()``  -> ``mimic3_amd/csrc/libmi355vits.so``  (hipcc, --offload-arch=gfx950; the product)
``build_emu()``  -> ``tests/emu/libmi355vits_emu.so``    (g++, -DMI355_EMU; CPU model of the same
                    sources for the ``-m "not gpu"`` tests — never loaded by the product path)

Both are rebuilt only when a source is newer than the library.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
EMU = os.path.join(ROOT, "tests", "emu")
SOURCES = ["kernels_conv.cpp", "kernels_mrf.cpp", "kernels_mrfp.cpp", "kernels_attn.cpp", "kernels_wn.cpp", "kernels_misc.cpp", "engine.cpp", "c_api.cpp"]
HIP_LIB = os.path.join(CSRC, "libmi355vits.so")
EMU_LIB = os.path.join(EMU, "libmi355vits_emu.so")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(extra=()):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".h"))]
    deps.append(os.path.join(ROOT, "include", "mi355vits.h"))
    deps.extend(extra)
    return deps


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def build_hip(force: bool = False, verbose_resources: bool = False) -> str:
    if not force and not _stale(HIP_LIB, _deps()):
        return HIP_LIB
    hipcc = find_hipcc()
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip",
           "-Wno-unused-result", "-I", os.path.join(ROOT, "include")]
    if verbose_resources:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", HIP_LIB + ".tmp"]
    _run(cmd)
    os.replace(HIP_LIB + ".tmp", HIP_LIB)
    return HIP_LIB


def build_emu(force: bool = False) -> str:
    extra = [os.path.join(EMU, "hip_emu.h"), os.path.join(EMU, "hip_emu_impl.cpp")]
    if not force and not _stale(EMU_LIB, _deps(extra)):
        return EMU_LIB
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-DMI355_EMU", "-Wno-psabi", "-Wno-unused-result",
           "-I", EMU, "-I", os.path.join(ROOT, "include")]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(EMU, "hip_emu_impl.cpp")]
    cmd += ["-o", EMU_LIB + ".tmp", "-lpthread"]
    _run(cmd)
    os.replace(EMU_LIB + ".tmp", EMU_LIB)
    return EMU_LIB


if __name__ == "__main__":
    which = sys.argv[1:] or ["hip", "emu"]
    if "hip" in which:
        print(build_hip(force="--force" in which, verbose_resources="--resources" in which))
    if "emu" in which:
        print(build_emu(force="--force" in which))
