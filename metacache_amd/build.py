"""Builds the in-tree native libraries.

    libmetacache_amd.so   the product: HIP kernels (gfx950) + C-ABI host code   [hipcc]
    bin/mcq               `metacache query` command line above the C ABI         [g++]

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container and on the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmetacache_amd.so")
SOURCES = ["kernels.hip", "gw_kernels.hip", "gw_sort.hip", "table_build.hip", "context.cpp", "dbfile.cpp", "dbload.cpp", "builder.hip", "partset.cpp", "keyshard.hip", "keyset.cpp", "devcache.cpp"]
HEADERS = ["kernels.h", "device_common.h", "context.h", "rccl_dl.h", "devcache.h", os.path.join(ROOT, "include", "metacache_amd.h")]
ARCH = "gfx950"
BINDIR = os.path.join(PKG, "bin")
MCQ = os.path.join(BINDIR, "mcq")


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X library cannot be built")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    hipcc = _hipcc()
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-I", os.path.join(ROOT, "include"),
             "-Wno-unused-result"] + os.environ.get("MC_HIPCC_FLAGS", "").split()
    procs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s).rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + flags + ["-c", s, "-o", o]
            if verbose:
                print("+", " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + " ".join(cmd))
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_cli(force=force, verbose=verbose)
    build_synth(force=force, verbose=verbose)
    build_gather_peak(force=force, verbose=verbose)
    build_slot_driver(force=force, verbose=verbose)
    return LIB


SYNTH_DIR = os.path.join(PKG, "synth")
SYNTH_LIB = os.path.join(LIBDIR, "libmcsynth.so")
SYNTH_CPU_LIB = os.path.join(LIBDIR, "libmcsynth_cpu.so")


def build_synth(force: bool = False, verbose: bool = False) -> tuple[str, str]:
    """The synthetic-workload generators (metacache_amd/synth): libmcsynth.so (HIP) and libmcsynth_cpu.so (C).
    Workload generation for bench.py / tests / tools only -- separate libraries, nothing of it is in the product."""
    os.makedirs(LIBDIR, exist_ok=True)
    spec = os.path.join(SYNTH_DIR, "synth_spec.h")
    hip = os.path.join(SYNTH_DIR, "synth.hip")
    cpu = os.path.join(SYNTH_DIR, "synth_cpu.c")
    if force or _stale(SYNTH_LIB, [hip, spec]):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-fPIC", "-shared", "-o", SYNTH_LIB, hip]
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if force or _stale(SYNTH_CPU_LIB, [cpu, spec]):
        cmd = ["gcc", "-O2", "-std=c99", "-fPIC", "-shared", "-o", SYNTH_CPU_LIB, cpu]
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SYNTH_LIB, SYNTH_CPU_LIB


GATHER_LIB = os.path.join(LIBDIR, "libmcgather.so")


def build_gather_peak(force: bool = False, verbose: bool = False) -> str:
    """tools/gather_peak.hip -> libmcgather.so: the random-access microbenchmark behind bench.py's second roofline (measurement tool)"""
    os.makedirs(LIBDIR, exist_ok=True)
    src = os.path.join(ROOT, "tools", "gather_peak.hip")
    if force or _stale(GATHER_LIB, [src]):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-fPIC", "-shared", "-Wno-unused-result", "-Wno-unused-value", "-o", GATHER_LIB, src]
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return GATHER_LIB


SLOTDRV_LIB = os.path.join(LIBDIR, "libmcslotdrv.so")


def build_slot_driver(force: bool = False, verbose: bool = False) -> str:
    """tools/slot_driver.cpp -> libmcslotdrv.so: host threads driving the slot API without an interpreter (measurement tool, tools/slot_path_bench.py)"""
    src = os.path.join(ROOT, "tools", "slot_driver.cpp")
    if force or _stale(SLOTDRV_LIB, [src, LIB, os.path.join(ROOT, "include", "metacache_amd.h")]):
        cmd = ["g++", "-std=c++14", "-O2", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), src, "-o", SLOTDRV_LIB,
               "-L", LIBDIR, "-lmetacache_amd", "-Wl,-rpath,$ORIGIN", "-pthread"]
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return SLOTDRV_LIB


def build_cli(force: bool = False, verbose: bool = False) -> str:
    """mcq: plain host C++14 linked against the C ABI only (rpath to the in-tree library)."""
    os.makedirs(BINDIR, exist_ok=True)
    src = os.path.join(CSRC, "mcq_main.cpp")
    if force or _stale(MCQ, [src, os.path.join(CSRC, "mcq_common.h"), os.path.join(CSRC, "mcq_build.h"), LIB,
                           os.path.join(ROOT, "include", "metacache_amd.h")]):
        cmd = ["g++", "-std=c++14", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", MCQ,
               "-L", LIBDIR, "-lmetacache_amd", "-lz", "-Wl,-rpath,$ORIGIN/../lib", "-pthread"]
        if verbose:
            print("+", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return MCQ


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
