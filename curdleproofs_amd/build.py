"""Builds libcpx.so (HIP kernels + host engine + C-ABI) for gfx950 with hipcc, in-tree.

    python -m curdleproofs_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting curdleproofs_amd/_lib/libcpx.so travels to the
GPU box with the repository snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_lib")
LIB = os.path.join(OUT_DIR, "libcpx.so")
ARCH = "gfx950"
SOURCES = ["kernels.hip", "late.hip", "protocol.hip", "round.hip", "engine.cpp", "engine_device.cpp", "whisk.cpp", "capi.cpp"]
# host side: x86-64-v3 + ADX (BMI2 mulx / andn / rorx: the Keccak permutation of the transcripts runs 1.7x faster, the
# 64-bit-limb Fr products use mulx); every host of an MI355X (EPYC 9005) has them
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-ffp-contract=off",
         "-Xarch_host", "-march=x86-64-v3", "-Xarch_host", "-madx"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the MI355X core cannot be built (there is no CPU fallback)")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "cpx.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines, verbose=False):
    """Builds an experimental variant libcpx_<name>.so with extra -D flags (A/B runs; select it at load
    time with the CPX_LIB environment variable)."""
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = _hipcc()
    out = os.path.join(OUT_DIR, "libcpx_%s.so" % name)
    cmd = [hipcc] + FLAGS + ["-D" + d for d in defines] + ["-shared", "-x", "hip"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out, "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    """Builds libcpx.so if any source is newer.  Concurrent callers (a test session and a manual build, several ranks starting at
    once) are serialised by a lock file, objects go to a directory of their own per build and the finished library is moved into
    place atomically: a reader never sees a half-written or mixed-generation libcpx.so."""
    import fcntl
    if not force and not _stale():
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():      # somebody else built it while we waited
            return LIB
        hipcc = _hipcc()
        work = os.path.join(OUT_DIR, "build.tmp")   # (fixed name: builds are serialised by the lock, and the output stays reproducible)
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(work)
        try:
            objs = []
            procs = []
            for src in SOURCES:
                obj = os.path.join(work, src + ".o")
                cmd = [hipcc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
                objs.append(obj)
            for src, p in procs:
                out, _ = p.communicate()
                if p.returncode != 0:
                    sys.stderr.write(out.decode())
                    raise RuntimeError("hipcc failed on " + src)
                if verbose and out:
                    print(out.decode())
            tmp_lib = os.path.join(work, "libcpx.so")
            subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", tmp_lib] + objs + ["-lpthread"])
            os.replace(tmp_lib, LIB)
        finally:
            shutil.rmtree(work, ignore_errors=True)
        build_selfcheck()
    return LIB


SELFCHECK = os.path.join(OUT_DIR, "quad_selfcheck")


def build_selfcheck():
    """The device-side self-check of the quad-cooperative point formulas (g1_28_quad.hpp against the one-lane formulas of
    g1_28.hpp, every special case), an executable that travels with the library; tests/test_gpu_parity.py runs it."""
    os.makedirs(OUT_DIR, exist_ok=True)
    src = os.path.join(HERE, "..", "scripts", "micro", "quad_micro.hip")
    subprocess.check_call([_hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-I", CSRC, src, "-o", SELFCHECK])
    return SELFCHECK


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
