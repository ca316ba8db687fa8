"""Build the native library.

``build_hip()``  -> ``mimic3_amd/csrc/libmi355vits.so``  (hipcc, --offload-arch=gfx950; the product: include/mi355vits.h only)
                    + ``libmi355vits_hooks.so`` (the same objects + csrc/lab_api.cpp: include/mi355vits_lab.h, tests and bench probe)
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
SOURCES = ["kernels_conv.cpp", "kernels_mrf.cpp", "kernels_mrfp.cpp", "kernels_mrfs.cpp", "kernels_rbc.cpp", "kernels_attn.cpp", "kernels_wn.cpp", "kernels_misc.cpp", "engine.cpp", "c_api.cpp"]
# csrc/lab_api.cpp = the hooks of include/mi355vits_lab.h (kernel unit tests, conv micro-benchmark, box probes): NOT in the product
# library; linked with the product's own objects into libmi355vits_hooks.so, and compiled into the lab build and the CPU model
HOOK_SOURCES = ["lab_api.cpp"]
PER_FILE_FLAGS = {}
LAB_FILE_FLAGS = {}  # per-file flags of the lab build's side of a running A/B (none at the moment)
# throw-away instrumented twins of the lab build (MI355_LAB_VARIANT=<name> python -m mimic3_amd.build lab -> libmi355vits_lab_<name>.so)
VARIANT_FILE_FLAGS = {"clk": {"kernels_mrfp.cpp": ["-DMRFP_CLOCKS"]}}  # shader-clock stamps inside k_mrf_p<32> (MI355VITS_MRFP_CLOCKS=1)
# sources the lab build compiles exactly as the product does (no -DMI355_LAB): k_mrf_p's in-loop ablation tests cost 15 % of its
# time, and it is now the REFERENCE side of the sweep kernels' A/B (its own ablations are in profiles/r03_mrf_experiments.txt)
LAB_AS_PRODUCT = {"kernels_mrfp.cpp"}
# sources of the lab build and the CPU model only (designs that measured equal or worse than what the product runs, kept for ONE round as
# the A/B of that statement; round 4's k_mrf_s1 was deleted in round 5 when k_rb_conv's staging-in-the-matrix-waves form took that place
# inside kernels_rbc.cpp)
LAB_ONLY = set()
HIP_LIB = os.path.join(CSRC, "libmi355vits.so")
HOOKS_LIB = os.path.join(CSRC, "libmi355vits_hooks.so")
EMU_LIB = os.path.join(EMU, "libmi355vits_emu.so")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(extra=()):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".h"))]
    deps.append(os.path.join(ROOT, "include", "mi355vits.h"))
    deps.append(os.path.join(ROOT, "include", "mi355vits_lab.h"))
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


def build_hip(force: bool = False, verbose_resources: bool = False, lab: bool = False) -> str:
    """One object per source (compiled in parallel, rebuilt only when the source or a header changed), then one link.
    ``lab=True`` builds ``libmi355vits_lab.so`` with -DMI355_LAB (timing / kernel-choice experiments for tools/; never
    loaded by the product)."""
    variant = os.environ.get("MI355_LAB_VARIANT", "") if lab else ""  # "plain": the lab build without LAB_FILE_FLAGS, as libmi355vits_lab_plain.so
    target = HIP_LIB.replace(".so", "_lab%s.so" % ("_" + variant if variant else "")) if lab else HIP_LIB
    hdrs = [d for d in _deps() if d.endswith(".h")]
    objdir = os.path.join(CSRC, "build", ("lab_" + variant if variant else "lab") if lab else "hip")
    os.makedirs(objdir, exist_ok=True)
    hipcc = find_hipcc()
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-result",
            "-I", os.path.join(ROOT, "include")]
    if verbose_resources:
        base.append("-Rpass-analysis=kernel-resource-usage")
    jobs = []
    srcs = [s for s in SOURCES if lab or s not in LAB_ONLY] + HOOK_SOURCES
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".cpp", ".o"))
        if force or verbose_resources or _stale(obj, [src] + hdrs):
            lab_flags = (["-DMI355_LAB"] if s not in LAB_AS_PRODUCT else []) + (VARIANT_FILE_FLAGS.get(variant, {}).get(s, []) if variant else LAB_FILE_FLAGS.get(s, [])) if lab else []
            jobs.append(base + PER_FILE_FLAGS.get(s, []) + lab_flags + ["-c", src, "-o", obj])
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(_run, jobs))
    objs = [os.path.join(objdir, s.replace(".cpp", ".o")) for s in srcs]
    # the product library: everything but the hooks; the same objects + the hooks = libmi355vits_hooks.so (test infrastructure)
    links = [(target, objs)] if lab else [(target, [o for o, s in zip(objs, srcs) if s not in HOOK_SOURCES]), (HOOKS_LIB, objs)]
    for tgt, ob in links:
        if jobs or _stale(tgt, ob):
            _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + ob + ["-o", tgt + ".tmp"])
            os.replace(tgt + ".tmp", tgt)
    return target


def build_emu(force: bool = False) -> str:
    """The CPU model: one object per source (in parallel, rebuilt only when the source or a header changed), then one link."""
    extra = [os.path.join(EMU, "hip_emu.h"), os.path.join(EMU, "hip_emu_impl.cpp")]
    hdrs = [d for d in _deps(extra) if d.endswith(".h")]
    objdir = os.path.join(EMU, "build")
    os.makedirs(objdir, exist_ok=True)
    cxx = os.environ.get("CXX", "g++")
    base = [cxx, "-O2", "-std=c++17", "-fPIC", "-DMI355_EMU", "-Wno-psabi", "-Wno-unused-result", "-I", EMU, "-I", os.path.join(ROOT, "include")]
    srcs = [os.path.join(CSRC, s) for s in SOURCES + HOOK_SOURCES] + [os.path.join(EMU, "hip_emu_impl.cpp")]
    objs = [os.path.join(objdir, os.path.basename(s).replace(".cpp", ".o")) for s in srcs]
    jobs = [base + ["-c", s, "-o", o] for s, o in zip(srcs, objs) if force or _stale(o, [s] + hdrs)]
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(_run, jobs))
    if jobs or _stale(EMU_LIB, objs):
        _run([cxx, "-shared", "-fPIC"] + objs + ["-o", EMU_LIB + ".tmp", "-lpthread"])
        os.replace(EMU_LIB + ".tmp", EMU_LIB)
    return EMU_LIB


if __name__ == "__main__":
    which = sys.argv[1:] or ["hip", "emu"]
    if "hip" in which:
        print(build_hip(force="--force" in which, verbose_resources="--resources" in which))
    if "lab" in which:
        print(build_hip(force="--force" in which, lab=True))
    if "emu" in which:
        print(build_emu(force="--force" in which))
