"""CPU checks of the drop-in boundary: the shared library builds, loads, exports every symbol include/cpx.h
declares, and fails loudly (no CPU fallback) when no HIP device is present.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from curdleproofs_amd.build import build
    build()
    import curdleproofs_amd as cpx
    return cpx.load_library()


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "cpx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cpx_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_library_agree(lib):
    import curdleproofs_amd as cpx
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libcpx.so does not export %s declared in include/cpx.h" % name
    assert sorted(cpx.EXPORTS) == declared, "python binding list and header differ"


def test_every_entry_point_cites_the_reference():
    hdr = open(os.path.join(ROOT, "include", "cpx.h")).read()
    for needle in ("util.rs:19-22", "util.rs:25-29", "inner_product_argument.rs:177-178", "grand_product_argument.rs:90-102",
                   "msm_accumulator.rs:22-68", "curdleproofs.rs:59", "curdleproofs.rs:197", "crs.rs:37-58",
                   "whisk.rs:144-179", "whisk.rs:106-130", "whisk.rs:228-263", "whisk.rs:183-226"):
        assert needle in hdr


def test_no_gpu_means_loud_failure(lib):
    import curdleproofs_amd as cpx
    if lib.cpx_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = ctypes.c_void_p()
    assert lib.cpx_ctx_create(0, ctypes.byref(h)) == cpx.CPX_ERR_HIP
    assert not h.value
    with pytest.raises(cpx.CpxError):
        cpx.Context(0)


def test_null_arguments_are_rejected_not_dereferenced(lib):
    import curdleproofs_amd as cpx
    assert lib.cpx_ctx_create(0, None) == cpx.CPX_ERR_ARG
    assert lib.cpx_g1_msm(None, None, None, 0, None) == cpx.CPX_ERR_ARG
    assert lib.cpx_batch_prove(None, None, None, None, None, None) == cpx.CPX_ERR_ARG
    assert lib.cpx_whisk_generate_shuffle_proof(None, None, None, None, None, None, None, None) == cpx.CPX_ERR_ARG
    assert lib.cpx_whisk_is_valid_tracker_proof(None, None, None, None, None) == cpx.CPX_ERR_ARG
    assert lib.cpx_batch_load_begin(None, 1, None, None, None, None, None) == cpx.CPX_ERR_ARG
    assert lib.cpx_batch_load_end(None) == cpx.CPX_ERR_ARG
    assert lib.cpx_proof_size(None) == 0
    assert isinstance(lib.cpx_last_error(None), bytes)   # NULL context: the calling thread's last cpx_ctx_create failure (or nothing)
    lib.cpx_ctx_destroy(None)


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "curdleproofs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in src and "liborc" not in src and "oracle/" not in src, f
