"""A compiled C program as the caller of the boundary: tests/cabi_smoke.c is built with gcc against include/cpx.h and linked
against libcpx.so (no ctypes in the call path).  CPU: it compiles and links.  GPU: it runs util::msm and an ell = 28
CurdleproofsProof::new / verify against inputs and expected outputs written by the oracle (README.md:76-118 recipe, seed 0)."""
import hashlib
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host_emul", "cabi_smoke")


def _build():
    from curdleproofs_amd.build import build, OUT_DIR
    build()
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi_smoke.c"), "-o", EXE,
           "-L", OUT_DIR, "-lcpx", "-Wl,-rpath," + OUT_DIR, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_c_caller_compiles_and_links_against_the_header():
    exe = _build()
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr        # starts, needs its input file


@pytest.mark.gpu
def test_c_caller_msm_prove_verify(orc, oracle_vectors, tmp_path):
    exe = _build()
    ell, n = 28, 32
    crs = orc.generate_crs_points(ell)
    inst = orc.make_instance(ell, 0, crs)
    assert hashlib.sha256(inst["proof"]).hexdigest() == oracle_vectors["28"]["proof_sha256"]   # the committed golden vector
    rng = orc.rng(2024)
    npts = 200
    bases, scalars = rng.g1_affine(npts), rng.fr(npts)
    want = orc.g1_compress_jac(orc.g1_msm(bases, scalars))
    blobs = [struct.pack("<Q", ell), crs, bases, scalars, want, inst["vec_R"], inst["vec_S"], inst["vec_T"], inst["vec_U"], inst["M"],
             struct.pack("<%dI" % ell, *inst["permutation"]), inst["k"], inst["vec_m_blinders"], inst["prover_rand"], inst["proof"], inst["verifier_rand"]]
    path = tmp_path / "cabi_smoke.in"
    with open(path, "wb") as f:
        for b in blobs:
            f.write(struct.pack("<Q", len(b)))
            f.write(b)
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cabi_smoke ok" in r.stdout, r.stdout + r.stderr
