"""A compiled C program as the caller of the boundary: tests/cabi_smoke.c is built with gcc against include/cpx.h and linked
against libcpx.so (no ctypes in the call path).  CPU: it compiles and links.  GPU: it runs util::msm and an ell = 28
CurdleproofsProof::new / verify against inputs and expected outputs written by the oracle (README.md:76-118 recipe, seed 0), then
cpx_accum_*, cpx_batch_verify_fused + cpx_g1_sum_jac and the four cpx_whisk_* calls on the reference's own known-answer vectors."""
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
def test_c_caller_msm_prove_verify(orc, oracle_vectors, whisk_kat, tmp_path):
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
    # second section: MsmAccumulator, the fused verifier, and the four Whisk entry points on the reference's own vectors
    # (tests/golden/whisk_kat.json: whisk.rs:401 and :455); inputs drawn as the reference's tests draw them (StdRng(0))
    an = 40
    V, x = rng.g1_affine(an), rng.fr(an)
    C = orc.g1_msm(V, x)
    blobs += [V, x, C, orc.g1_add_jac(C, C), rng.fr(2), rng.fr(12)]
    r0 = orc.rng(0)
    gen = orc.g1_generator()
    tk, tr = r0.fr(1), r0.fr(1)                                    # whisk.rs:383-385: k, then WhiskTracker::from_k draws r
    rG = orc.g1_scale(gen, tr)
    tracker = orc.g1_compress(rG) + orc.g1_compress(orc.g1_scale(rG, tk))
    blobs += [tracker, tk, r0.fr(1), orc.g1_compress(orc.g1_scale(gen, tk)), bytes.fromhex(whisk_kat["tracker_proof"])]
    ell2 = 124
    r0 = orc.rng(0)
    pre = b""
    for _ in range(ell2):                                          # whisk.rs:416-424 generate_shuffle_trackers
        kk, rr = r0.fr(1), r0.fr(1)
        rG = orc.g1_scale(gen, rr)
        pre += orc.g1_compress(rG) + orc.g1_compress(orc.g1_scale(rG, kk))
    perm2 = r0.shuffle(ell2)
    blobs += [struct.pack("<Q", ell2), orc.generate_crs_points(ell2), pre, struct.pack("<%dI" % ell2, *perm2), r0.fr(1), r0.fr(4), r0.fr(3 * (ell2 + 4) + 9),
              bytes.fromhex(whisk_kat["whisk_shuffle_proof_ell124"]), r0.fr(8)]
    path = tmp_path / "cabi_smoke.in"
    with open(path, "wb") as f:
        for b in blobs:
            f.write(struct.pack("<Q", len(b)))
            f.write(b)
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cabi_smoke ok" in r.stdout and "second section ok" in r.stdout, r.stdout + r.stderr
