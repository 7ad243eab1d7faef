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
