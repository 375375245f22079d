"""Build helpers: compile the gfx950 library and the host programs on top of its C ABI.

Everything is built IN-TREE (the .so / binaries travel to the GPU box with the source snapshot):
  necat_amd/csrc/libnecat_hip.so   hipcc --offload-arch=gfx950   (the product)
  necat_amd/csrc/oc2pmov, oc2pm    host programs on top of the C ABI
The test oracle (oracle/) is built by tests/oracle_build.py - this package knows nothing about it.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libnecat_hip.so")
LIB_XCHECK = os.path.join(CSRC, "libnecat_hip_xcheck.so")      # the same sources + the retired kernel families (-DNECAT_BUILD_CROSSCHECK): what the tests' alternative-path cases load
OC2PMOV = os.path.join(CSRC, "oc2pmov")
OC2PM = os.path.join(CSRC, "oc2pm")
OC2MKDB = os.path.join(CSRC, "oc2mkdb")
OC2PCAN = os.path.join(CSRC, "oc2pcan")
OC2CNS = os.path.join(CSRC, "oc2cns")
OC2RM = os.path.join(CSRC, "oc2rm_worker")
OC2ASMPM = os.path.join(CSRC, "oc2asmpm")

HIP_SOURCES = ["necat_hip.hip"]          # one translation unit; its stages are the stage_*.inl files it includes


def _hip_deps():
    """every header next to the translation unit + the public header: editing any of them rebuilds the library"""
    import glob
    return HIP_SOURCES + sorted(os.path.basename(h) for h in glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "stage_*.inl"))) + \
        sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))


# -disable-promote-alloca-to-lds: the compiler otherwise parks k_seed_eval's small per-lane arrays in LDS (3.8 KB per wave), and that
# kernel's occupancy is bounded by its LDS (no other kernel changes)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-mllvm", "-disable-promote-alloca-to-lds"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built (there is no CPU fallback)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for d in deps:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build step failed: %s" % " ".join(cmd))
    return r.stdout


def build_hip(force: bool = False) -> str:
    if force or _stale(LIB, _hip_deps()):
        _run([_hipcc()] + HIPCC_FLAGS + ["-shared", "-o", LIB] + HIP_SOURCES, cwd=CSRC)
    return LIB


def build_xcheck(force: bool = False) -> str:
    """the cross-check build (necat_hip.hip, NECAT_BUILD_CROSSCHECK): the product's code + the kernel families its default paths replaced; test infrastructure -
    no program links it"""
    if force or _stale(LIB_XCHECK, _hip_deps()):
        _run([_hipcc()] + HIPCC_FLAGS + ["-DNECAT_BUILD_CROSSCHECK", "-shared", "-o", LIB_XCHECK] + HIP_SOURCES, cwd=CSRC)
    return LIB_XCHECK


def build_cli(force: bool = False):
    build_hip()
    if force or _stale(OC2PMOV, ["oc2pmov_main.cpp", "pm_job.h", "host_fmt.h", "host_io.h", LIB]):
        _run([_hipcc(), "-O2", "-std=c++17", "-o", OC2PMOV, "oc2pmov_main.cpp", "-L" + CSRC, "-lnecat_hip",
              "-Wl,-rpath,$ORIGIN", "-lpthread"], cwd=CSRC)
    if force or _stale(OC2PM, ["oc2pm_main.cpp", "pm_job.h", "host_fmt.h", "host_io.h", LIB]):       # one resident worker per GPU runs the volume jobs itself
        _run([_hipcc(), "-O2", "-std=c++17", "-o", OC2PM, "oc2pm_main.cpp", "-L" + CSRC, "-lnecat_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"], cwd=CSRC)
    if force or _stale(OC2MKDB, ["oc2mkdb_main.cpp"]):          # host-only drop-in of the volume writer
        _run([shutil.which("g++") or "g++", "-O2", "-std=c++17", "-o", OC2MKDB, "oc2mkdb_main.cpp", "-lz", "-ldl"], cwd=CSRC)
    if force or _stale(OC2PCAN, ["oc2pcan_main.cpp"]):          # host-only drop-in of the candidate partitioner
        _run([shutil.which("g++") or "g++", "-O2", "-std=c++17", "-o", OC2PCAN, "oc2pcan_main.cpp"], cwd=CSRC)
    if force or _stale(OC2CNS, ["oc2cns_main.cpp", "cns_consensus.h", "host_io.h", LIB]):   # consensus stage: GPU extension loop + host consensus
        _run([_hipcc(), "-O2", "-std=c++17", "-o", OC2CNS, "oc2cns_main.cpp", "-L" + CSRC, "-lnecat_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"], cwd=CSRC)
    if force or _stale(OC2RM, ["oc2rm_worker_main.cpp", "pm_job.h", "host_fmt.h", "host_io.h", LIB]):   # reads against a reference (second half of SURVEY 8f.4)
        _run([_hipcc(), "-O2", "-std=c++17", "-o", OC2RM, "oc2rm_worker_main.cpp", "-L" + CSRC, "-lnecat_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"], cwd=CSRC)
    if force or _stale(OC2ASMPM, ["oc2asmpm_main.cpp", "asm_job.h", "asm_core.h", "rescue.h", "host_fmt.h", "host_io.h", LIB]):   # overlapper of corrected reads
        _run([_hipcc(), "-O2", "-std=c++17", "-ffp-contract=off", "-o", OC2ASMPM, "oc2asmpm_main.cpp", "-L" + CSRC, "-lnecat_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"], cwd=CSRC)
    return OC2PMOV, OC2PM


def build_all(force: bool = False):
    build_hip(force)
    build_xcheck(force)
    build_cli(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
    print("built:", LIB)
