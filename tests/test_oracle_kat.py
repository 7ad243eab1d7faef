"""Pins the CPU oracle (oracle/) to every fixed output the reference's own tests hold
(SURVEY.md §8c): whisk.rs:355-361, :363-368, :381-402, :416-456, plus the published Merlin
vector and Keccak (via hashlib's SHA3).  CPU only."""
import hashlib
import os
import struct

import pytest

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def test_keccak_against_hashlib_sha3(orc):
    # SHA3-256 of a short message = one Keccak-f[1600] call on the padded block (rate 136)
    msg = b"curdleproofs oracle keccak check"
    st = bytearray(200)
    st[: len(msg)] = msg
    st[len(msg)] ^= 0x06
    st[135] ^= 0x80
    out = orc.keccak_f1600(bytes(st))
    assert out[:32] == hashlib.sha3_256(msg).digest()


def test_merlin_published_vector(orc):
    assert orc.merlin_test_vector().hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_stdrng_checkpoints(orc):
    rng = orc.rng(0)
    assert rng.key().hex() == "ecf273f981b5cd4587f0467306ad6cadd0d0a3e33317e767f29bea72d78a7dfe"
    assert [rng.u32() for _ in range(4)] == [0xCD2C6F7F, 0xBB2A3FB2, 0x8E27697B, 0xC6017C94]


def test_fr_rand_is_montgomery_limbs(orc):
    # SURVEY §8c checkpoint: first Fr::rand of StdRng(0), canonical LE bytes
    fr = orc.rng(0).fr(1)
    assert orc.fr_to_canonical_bytes(fr).hex() == "7a497eeda6d8ed1f38c9324427a251a98a215f13d93c3a138a727b5c27019269"


def test_fr_bytes_roundtrip_kat(orc, whisk_kat):
    # whisk.rs:355-361 serde_fr_rand
    k_bytes = bytes.fromhex(whisk_kat["fr_le_bytes"])
    k = orc.fr_from_le_bytes_mod_order(k_bytes)
    assert orc.fr_to_canonical_bytes(k) == k_bytes
    assert orc.fr_from_canonical_bytes(k_bytes) == k


def test_g1_generator_roundtrip_kat(orc, whisk_kat):
    # whisk.rs:363-368 serde_g1_roundtrip
    gen = bytes.fromhex(whisk_kat["g1_generator_compressed"])
    p = orc.g1_decompress(gen)
    assert p == orc.g1_generator()
    assert orc.g1_compress(p) == gen
    assert orc.g1_on_curve(p) and orc.g1_in_subgroup(p)


def test_field_arithmetic_against_python_ints(orc):
    # independent check of the Montgomery code with Python big ints
    rnd = os.urandom
    for mod, nbytes, mul, inv in ((P, 48, orc.fp_mul, orc.fp_inv), (R, 32, orc.fr_mul, orc.fr_inv)):
        Rm = 1 << (8 * nbytes)
        a = [int.from_bytes(rnd(nbytes), "little") % mod for _ in range(16)] + [0, 1, mod - 1]
        b = [int.from_bytes(rnd(nbytes), "little") % mod for _ in range(16)] + [mod - 1, mod - 1, mod - 1]
        ab = b"".join(x.to_bytes(nbytes, "little") for x in a)
        bb = b"".join(x.to_bytes(nbytes, "little") for x in b)
        got = mul(ab, bb)
        Rinv = pow(Rm, -1, mod)
        for i, (x, y) in enumerate(zip(a, b)):
            assert int.from_bytes(got[i * nbytes:(i + 1) * nbytes], "little") == x * y * Rinv % mod
        gi = inv(ab)
        for i, x in enumerate(a):
            if x:
                # inverse in Montgomery form: (x R^-1)^-1 * R = R^2 / x
                assert int.from_bytes(gi[i * nbytes:(i + 1) * nbytes], "little") == pow(x, -1, mod) * Rm * Rm % mod


def test_crs_checkpoint(orc):
    # generate_crs(124).vec_G[0] compressed starts ac5f2023...
    crs = orc.generate_crs_points(4)   # same StdRng(0) stream prefix for any ell
    assert orc.g1_compress(crs[:96]).hex().startswith("ac5f20234d022490c77c18f9a9ec8458")


def test_shuffle_checkpoint(orc):
    # the permutation of KAT 4 is drawn after 248 tracker scalars (whisk.rs:418-424, :152-153)
    rng = orc.rng(0)
    rng.fr(248)
    assert rng.shuffle(124)[:8] == [30, 94, 39, 35, 69, 3, 52, 115]


def test_tracker_proof_kat(orc, whisk_kat):
    # whisk.rs:381-402
    valid, proof = orc.kat_tracker_proof()
    assert valid == 1
    assert proof.hex() == whisk_kat["tracker_proof"]


def test_whisk_shuffle_proof_kat(orc, whisk_kat):
    # whisk.rs:416-456: 4496 bytes = M || CurdleproofsProof::serialize at ell = 124
    valid, proof, pre, post = orc.kat_shuffle_proof(124)
    assert valid == 1
    assert len(proof) == 4496
    assert proof.hex() == whisk_kat["whisk_shuffle_proof_ell124"]


def test_oracle_whisk_entry_points_with_explicit_draws(orc, whisk_kat):
    """The explicit-draw forms the GPU whisk tests compare against reproduce the reference KATs too (whisk.rs:381-456)."""
    rng = orc.rng(0)
    k, r = rng.fr(1), rng.fr(1)
    gen = orc.g1_generator()
    rG = orc.g1_scale(gen, r)
    tracker = orc.g1_compress(rG) + orc.g1_compress(orc.g1_scale(rG, k))
    proof = orc.generate_whisk_tracker_proof(tracker, k, rng.fr(1))
    assert proof.hex() == whisk_kat["tracker_proof"]
    kc = orc.g1_compress(orc.g1_scale(gen, k))
    assert orc.is_valid_whisk_tracker_proof(tracker, kc, proof) == 1
    assert orc.is_valid_whisk_tracker_proof(tracker, orc.g1_compress(gen), proof) == 0
    assert orc.is_valid_whisk_tracker_proof(tracker, kc, proof[:96] + b"\xff" * 32) == -1
    v, sproof, pre, post = orc.kat_shuffle_proof(124)
    crs = orc.generate_crs_points(124)
    vr = orc.rng(5).fr(8)
    assert v == 1 and orc.is_valid_whisk_shuffle_proof(124, crs, pre, post, sproof, vr) == 1
    assert orc.is_valid_whisk_shuffle_proof(124, crs, post, pre, sproof, vr) == 0
    assert orc.is_valid_whisk_shuffle_proof(124, crs, pre, post, bytes(48) + sproof[48:], vr) == -1


def test_sha256_matches_hashlib(orc):
    import hashlib
    for msg in (b"", b"abc", b"nankokita_no_kakurenbo" + (7).to_bytes(8, "big"), bytes(range(256)) * 3, b"x" * 55, b"y" * 56, b"z" * 64):
        assert orc.sha256(msg) == hashlib.sha256(msg).digest()


def test_hash_to_curve_points_match_an_independent_big_integer_restatement(orc):
    """/root/reference/tests/crs.rs:13-52 with the reference's seed string.  The reference holds no vector for these points (the test
    prints them), so the oracle's restatement is pinned here against a second, independent one over Python integers: SHA-256 from
    hashlib, x from the four big-endian words as little-endian limbs, y = the smaller square root of x^3 + 4, times the full cofactor."""
    import hashlib
    p = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    h = 0x396c8c005555e1568c00aaab0000aaab
    seed = b"nankokita_no_kakurenbo"

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return x3, (lam * (x1 - x3) - y1) % p

    def mul(P, k):
        R = None
        for bit in bin(k)[2:]:
            R = add(R, R)
            if bit == "1":
                R = add(R, P)
        return R

    want, inc = [], 0
    while len(want) < 5:
        d = hashlib.sha256(seed + inc.to_bytes(8, "big")).digest()
        x = sum(int.from_bytes(d[8 * i:8 * i + 8], "big") << (64 * i) for i in range(4))
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        inc += 1
        if y * y % p != rhs:
            continue
        want.append(mul((x, min(y, p - y)), h))
    got, got_inc = orc.hash_to_curve_points(5, seed.decode())
    assert got_inc == inc
    R = 1 << 384
    for i, (x, y) in enumerate(want):
        rec = got[96 * i:96 * (i + 1)]
        gx = int.from_bytes(rec[:48], "little") * pow(R, -1, p) % p      # wire = Montgomery limbs
        gy = int.from_bytes(rec[48:], "little") * pow(R, -1, p) % p
        assert (gx, gy) == (x, y), i
        assert orc.g1_on_curve(rec) and orc.g1_in_subgroup(rec)


def test_fast_build_of_the_cpu_baseline_reproduces_the_known_answers(whisk_kat, oracle_vectors):
    """bench.py's `cpu_baseline` times the ORC_FAST build of the oracle (oracle/Makefile: -march=x86-64-v3 -madx, the Montgomery product of
    ark-ff's `asm` feature on mulx / adcx / adox, oracle/mont_asm.h).  A baseline that computes something else is no baseline: the same
    sources, the same known answers — the reference's tracker proof and its 4496-byte shuffle proof (whisk.rs:381-456), Merlin's vector,
    the field products against Python integers, and the ell = 28 README proof against the portable build's bytes."""
    from tests.oracle_lib import Oracle, fast_build_runs_here, oracle_build_flags
    if not fast_build_runs_here():
        pytest.skip("this CPU has no BMI2 + ADX")
    assert "-DORC_FAST" in oracle_build_flags(fast=True) and "-DORC_FAST" not in oracle_build_flags(fast=False)
    fast, port = Oracle(fast=True), Oracle()
    assert fast.merlin_test_vector() == port.merlin_test_vector()
    test_field_arithmetic_against_python_ints(fast)
    test_tracker_proof_kat(fast, whisk_kat)
    test_whisk_shuffle_proof_kat(fast, whisk_kat)
    crs = port.generate_crs_points(28)
    assert fast.generate_crs_points(28) == crs
    a, b = fast.make_instance(28, 0, crs), port.make_instance(28, 0, crs)
    assert a["proof"] == b["proof"] and a["vec_T"] == b["vec_T"] and a["M"] == b["M"]
    ok, tp, tv = fast.time_prove_verify(a, crs, 1)
    assert ok == 1 and tp > 0 and tv > 0


def test_infinity_flag_encodings_in_both_reading_modes(orc):
    """oracle/g1.h g1_decompress: the ark-bls12-381 ^0.4 reading (default: compression + infinity flag set = the identity, whatever else the
    48 bytes hold) and the strict one (only 0xc0 || 0^47) — the two behaviours the product's option strict_infinity selects between."""
    canon = bytes([0xc0]) + bytes(47)
    odd = [bytes([0xe0]) + bytes(47), bytes([0xc0]) + bytes(46) + b"\x01", b"\xff" * 48]
    try:
        for strict in (False, True):
            orc.set_strict_infinity(strict)
            assert orc.g1_decompress(canon) == bytes(96)
            for enc in odd:
                if strict:
                    with pytest.raises(ValueError):
                        orc.g1_decompress(enc)
                else:
                    assert orc.g1_decompress(enc) == bytes(96)
            with pytest.raises(ValueError):
                orc.g1_decompress(bytes([0x40]) + bytes(47))   # no compression flag: an error in both modes
    finally:
        orc.set_strict_infinity(False)


def test_committed_mont_asm_header_is_the_generators_output():
    """oracle/mont_asm.h (the mulx / adcx / adox product of the fast CPU-baseline build) is generated by oracle/gen_mont_asm.py and
    committed; the Makefile no longer regenerates it by mtime, so the two are held together here."""
    import os
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    out = subprocess.run([sys.executable, os.path.join(here, "gen_mont_asm.py")], capture_output=True, text=True, timeout=120, cwd=here)
    assert out.returncode == 0, out.stderr[-500:]
    with open(os.path.join(here, "mont_asm.h")) as f:
        assert f.read() == out.stdout
