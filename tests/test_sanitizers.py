"""Sanitizer builds of the product's host-side code: ThreadSanitizer and AddressSanitizer + UndefinedBehaviorSanitizer
over host_threads.hpp (worker pool, spin team: several engine contexts side by side), host_math.hpp / strobe.hpp (the host-driven
transcript) and the host/device-portable math headers, driven by tests/host_emul/sanitize_main.cpp.  Recipe: tests/host_emul/sanitize.sh;
the clean log of the round is committed as profiles/r03_sanitizers.txt."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_code_is_clean_under_tsan_asan_ubsan(tmp_path):
    r = subprocess.run(["sh", os.path.join(ROOT, "tests", "host_emul", "sanitize.sh"), str(tmp_path)], capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert out.count("sanitize driver: all checks passed") == 2 and "== clean" in out
    for bad in ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "runtime error:", "LeakSanitizer"):
        assert bad not in out, out[-4000:]


def _engine_input(orc, path):
    """instances and expected proofs for tests/host_emul/engine_sanitize_main.cpp: two CRS (ell = 28, 60), four instances each"""
    import struct
    blobs = [struct.pack("<Q", 2)]
    for ell in (28, 60):
        crs = orc.generate_crs_points(ell)
        blobs += [struct.pack("<Q", ell), crs, struct.pack("<Q", 4)]
        for seed in range(4):
            i = orc.make_instance(ell, 700 + seed, crs)
            blobs += [i["vec_R"], i["vec_S"], i["vec_T"], i["vec_U"], i["M"], struct.pack("<%dI" % ell, *i["permutation"]), i["k"], i["vec_m_blinders"],
                      i["prover_rand"], i["verifier_rand"], i["proof"]]
    with open(path, "wb") as f:
        for b in blobs:
            f.write(struct.pack("<Q", len(b)))
            f.write(b)


@pytest.mark.gpu
@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_engine_host_code_under_sanitizers_on_the_gpu(orc, tmp_path, san):
    """The engine's own host code — engine.cpp, engine_device.cpp (streams, events, plan caches, the shared registry of CRS tables), capi.cpp,
    whisk.cpp and the launchers — compiled with AddressSanitizer (+ leak detection) resp. ThreadSanitizer and run ON THE GPU
    (scripts/sanitize_engine.sh: device code unchanged; tests/host_emul/engine_sanitize_main.cpp: a lone proof, device-resident batches with fused
    and unfused rounds, option changes, four contexts switching CRS on four threads, a small batch beside GPU-filling ones).  Every proof is
    compared with the oracle's bytes inside the driver; the run must end without a sanitizer report of the product's code.  The HIP runtime
    is not instrumented: reports whose stacks lie entirely inside libamdhip64 / libhsa-runtime64 are filtered by the suppression files."""
    exe = os.path.join(ROOT, "curdleproofs_amd", "_lib", "engine_" + san)
    if not os.path.exists(exe):
        r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "sanitize_engine.sh"), san], capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    inp = tmp_path / "engine_san.in"
    _engine_input(orc, inp)
    supp = os.path.join(ROOT, "tests", "host_emul", "%s.supp" % san)
    env = dict(os.environ)
    if san == "asan":
        env["ASAN_OPTIONS"] = "detect_leaks=1:protect_shadow_gap=0:alloc_dealloc_mismatch=1"
        env["LSAN_OPTIONS"] = "suppressions=%s:print_suppressions=0" % supp
    else:
        env["TSAN_OPTIONS"] = "halt_on_error=0:second_deadlock_stack=1:suppressions=%s:ignore_noninstrumented_modules=1" % supp
    r = subprocess.run([exe, str(inp)], capture_output=True, text=True, timeout=1500, env=env)
    out = r.stdout + r.stderr
    log = os.path.join(ROOT, "gpurun_out", "prof")
    os.makedirs(log, exist_ok=True)
    with open(os.path.join(log, "sanitizers_engine_%s.txt" % san), "w") as f:
        f.write("$ %s %s   (exit code %d)\n%s" % (os.path.basename(exe), " ".join("%s=%s" % (k, env[k]) for k in env if k.endswith("SAN_OPTIONS")), r.returncode, out))
    assert "engine_san ok" in r.stdout, out[-6000:]
    for bad in ("WARNING: ThreadSanitizer", "ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:"):
        assert bad not in out, out[-8000:]
    assert r.returncode == 0, out[-4000:]
