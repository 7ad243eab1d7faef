"""CPU checks of the host-side Python mirror that need no GPU: scalar byte helpers of whisk.py (whisk.rs:333-345), argument
validation that happens before any device call, bench.py's cgroup-aware core count."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def test_whisk_scalar_byte_helpers(whisk_kat, orc):
    from curdleproofs_amd import whisk, params
    kb = bytes.fromhex(whisk_kat["fr_le_bytes"])                       # whisk.rs:355-361 serde_fr_rand
    k = whisk.from_bytes_fr(kb)
    assert whisk.to_bytes_fr(k) == kb
    assert k == orc.fr_from_canonical_bytes(kb)                        # the wire form the oracle (arkworks' Montgomery limbs) uses
    big = (R + 5).to_bytes(32, "little")                               # from_le_bytes_mod_order reduces
    assert whisk.to_bytes_fr(whisk.from_bytes_fr(big)) == (5).to_bytes(32, "little")
    assert params.fr_from_wire(params.fr_to_wire(R - 1)) == R - 1
    assert len(whisk.rand_scalar()) == 32 and whisk.rand_scalar() != whisk.rand_scalar()
    z = params.random_fr_wire(None, 64)
    assert all(0 < params.fr_from_wire(z[32 * i:32 * i + 32]) < R for i in range(64))


def test_whisk_tracker_shape_checks():
    from curdleproofs_amd import whisk
    t = whisk.WhiskTracker(b"\x01" * 48, b"\x02" * 48)
    assert t.to_bytes() == b"\x01" * 48 + b"\x02" * 48 and t == whisk.WhiskTracker(b"\x01" * 48, b"\x02" * 48)
    with pytest.raises(ValueError):
        whisk.WhiskTracker(b"\x01" * 47, b"\x02" * 48)
    assert whisk.TRACKER_PROOF_SIZE == 128 and whisk.G1POINT_SIZE == 48 and whisk.FIELD_ELEMENT_SIZE == 32   # whisk.rs:21-25


def test_util_helpers_without_device():
    from curdleproofs_amd import util
    assert util.get_permutation(list("abcd"), [2, 0, 3, 1]) == list("cadb")                    # util.rs:76-78
    assert util.inner_product([1, 2, 3], [4, 5, 6]) == 32                                      # util.rs:66-73
    with pytest.raises(AssertionError):
        util.inner_product([1], [1, 2])
    assert util.get_verification_scalars_bitstring(8, 3)[5] == [0, 2]                          # util.rs:39-64: 5 = 0b101


def test_effective_cores_is_positive_and_bounded():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
