"""CPU checks of the PRODUCT's portable math (curdleproofs_amd/csrc/*.hpp + host_math.hpp), compiled
for the host by tests/host_emul/emul.cpp, against the oracle.  The emulation library is test-only: it
mirrors what each kernel computes per lane so that arithmetic / recoding bugs surface without a GPU."""
import ctypes
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_emul", "emul.cpp")
LIB = os.path.join(HERE, "host_emul", "_emul.so")
CSRC = os.path.join(HERE, "..", "curdleproofs_amd", "csrc")
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
FR, AFF, JAC = 32, 96, 144


@pytest.fixture(scope="module")
def emul():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("mont32.hpp", "g1.hpp", "recode.hpp", "host_math.hpp", "fp28.hpp", "g1_28.hpp", "modinv30.hpp", "strobe.hpp", "quad_program.hpp", "host_threads.hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", LIB, SRC])
    L = ctypes.CDLL(LIB)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    L.emul_fp_mul.argtypes = [vp, vp, vp, sz]
    L.emul_fr_mul.argtypes = [vp, vp, vp, sz]
    L.emul_host_fr_mul.argtypes = [vp, vp, vp, sz]
    L.emul_host_fr_inv_check.argtypes = [vp, sz]
    L.emul_modinv30_check.argtypes = [vp, sz, vp]
    L.emul_glv_split.argtypes = [vp, sz, vp]
    L.emul_fr_inv_divsteps.argtypes = [vp, sz, vp]
    L.emul_fr_inv_divsteps.restype = None
    L.emul_msm_endo.argtypes = [vp, vp, sz, ctypes.c_int, vp]
    L.emul_msm_endo.restype = None
    L.emul_glv_split.restype = None
    L.emul_msm.argtypes = [vp, vp, sz, vp]
    L.emul_smul.argtypes = [vp, vp, vp, vp]
    L.emul_challenges.argtypes = [vp, sz, vp]
    L.emul_f28_mul.argtypes = [vp, vp, vp, sz]
    L.emul_f28_mulsub.argtypes = [vp, vp, vp, vp, vp, sz, ctypes.c_int]
    L.emul_f28_msm.argtypes = [vp, vp, sz, vp, vp]
    L.emul_f28_xyzz_sum.argtypes = [vp, sz, vp, vp, vp]
    L.emul_f28_xyzz_full.argtypes = [vp, sz, vp, vp, vp]
    L.emul_quad_program.argtypes = [ctypes.c_int, vp, vp]
    L.emul_quad_program.restype = None
    L.emul_thread_helpers.argtypes = [ctypes.c_int, sz, sz, sz, ctypes.c_long]
    L.emul_thread_helpers.restype = ctypes.c_int
    L.emul_fix_digits.argtypes = [ctypes.c_int, vp, sz, vp]
    L.emul_signed_nibbles.argtypes = [vp, sz, vp]
    L.emul_signed_nibbles.restype = None
    L.emul_late_uniform.argtypes = [vp, vp, sz, ctypes.c_int, vp]
    L.emul_late_uniform.restype = None
    L.emul_late_msm.argtypes = [vp, vp, sz, ctypes.c_int, vp]
    L.emul_late_msm.restype = None
    return L


def _b(x):
    return (ctypes.c_uint8 * len(x)).from_buffer_copy(x)


def _o(n):
    return (ctypes.c_uint8 * n)()


def test_limb_arithmetic_matches_oracle_and_python(emul, orc):
    rnd = os.urandom
    vals = [int.from_bytes(rnd(48), "little") % P for _ in range(64)] + [0, 1, P - 1, P - 2, (1 << 380), (1 << 381) - 1 - (1 << 380)]
    a = b"".join(v.to_bytes(48, "little") for v in vals)
    b = b"".join(v.to_bytes(48, "little") for v in reversed(vals))
    o = _o(len(a))
    emul.emul_fp_mul(_b(a), _b(b), o, len(vals))
    assert bytes(o) == orc.fp_mul(a, b)
    for x, y in zip(vals[:20] + vals[-6:], list(reversed(vals))[:20] + vals[-6:]):
        r = _o(4 * 48)
        emul.emul_fp_addsubnegdbl(_b(x.to_bytes(48, "little")), _b(y.to_bytes(48, "little")), r)
        got = [int.from_bytes(bytes(r)[i * 48:(i + 1) * 48], "little") for i in range(4)]
        assert got == [(x + y) % P, (x - y) % P, (-x) % P, (2 * x) % P]
    x = vals[3].to_bytes(48, "little")
    r = _o(48)
    emul.emul_fp_inv(_b(x), r)
    assert bytes(r) == orc.fp_inv(x)
    s = orc.rng(5).fr(40)
    t = orc.rng(6).fr(40)
    o = _o(len(s))
    emul.emul_fr_mul(_b(s), _b(t), o, 40)
    assert bytes(o) == orc.fr_mul(s, t)
    r = _o(32)
    emul.emul_fr_inv(_b(s[:32]), r)
    assert bytes(r) == orc.fr_inv(s[:32])


@pytest.mark.parametrize("n", [1, 2, 17, 64])
def test_msm_kernel_algorithm(emul, orc, n):
    rng = orc.rng(40 + n)
    bases = bytearray(rng.g1_affine(n))
    scalars = bytearray(rng.fr(n))
    if n >= 17:
        bases[0:AFF] = bytes(AFF)                                   # identity base
        bases[3 * AFF:4 * AFF] = bases[2 * AFF:3 * AFF]             # duplicate base + same scalar: P + P in a bucket
        scalars[3 * FR:4 * FR] = scalars[2 * FR:3 * FR]
        scalars[5 * FR:6 * FR] = bytes(FR)                          # zero scalar
        rm1 = (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 - 1).to_bytes(32, "little")
        scalars[6 * FR:7 * FR] = orc.fr_from_canonical_bytes(rm1)   # r - 1: top digits + carries
    o = _o(JAC)
    emul.emul_msm(_b(bytes(bases)), _b(bytes(scalars)), n, o)
    assert orc.g1_compress_jac(bytes(o)) == orc.g1_compress_jac(orc.g1_msm(bytes(bases), bytes(scalars), naive=True))


def test_smul_kernel_algorithm(emul, orc):
    rng = orc.rng(77)
    for i in range(6):
        A, M, s = rng.g1_affine(1), rng.g1_affine(1), rng.fr(1)
        if i == 1:
            A = bytes(AFF)
        if i == 2:
            M = bytes(AFF)
        if i == 3:
            s = orc.fr_from_u64(0)
        if i == 4:
            s = orc.fr_from_canonical_bytes((0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 - 1).to_bytes(32, "little"))
        o = _o(JAC)
        emul.emul_smul(_b(A), _b(M), _b(s), o)
        a = _o(AFF)
        emul.emul_to_affine(o, a)
        assert bytes(a) == orc.g1_fold(A, M, s)
        c = _o(48)
        emul.emul_compress(a, c)
        assert bytes(c) == orc.g1_compress(bytes(a))
        o2 = _o(JAC)
        emul.emul_smul(None, _b(M), _b(s), o2)
        emul.emul_to_affine(o2, a)
        assert bytes(a) == orc.g1_scale(M, s)


def test_engine_transcript(emul, orc):
    o = _o(32)
    emul.emul_merlin_vector(o)
    assert bytes(o).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    for ln in (0, 1, 47, 165, 166, 167, 400, 12104):
        msg = bytes((i * 7 + ln) & 0xFF for i in range(ln))
        o = _o(64)
        emul.emul_challenges(_b(msg) if ln else None, ln, o)
        assert bytes(o) == orc.challenges(msg)
    st = os.urandom(200)
    b = _b(st)
    emul.emul_keccak(b)
    assert bytes(b) == orc.keccak_f1600(st)


R_MONT = 1 << 384


def _to_mont(x):
    return (x * R_MONT % P).to_bytes(48, "little")


def _from_mont(b):
    return int.from_bytes(b, "little") * pow(R_MONT, -1, P) % P


def test_f28_field_matches_oracle(emul, orc):
    """The signed 28-bit-limb lazy field: conversions, products, lazy add/sub/shift chains, inversion."""
    vals = [int.from_bytes(os.urandom(48), "little") % P for _ in range(200)] + [0, 1, P - 1, P - 2, 1 << 380, (P - 1) // 2]
    a = b"".join(v.to_bytes(48, "little") for v in vals)
    b = b"".join(v.to_bytes(48, "little") for v in reversed(vals))
    o = _o(len(a))
    emul.emul_f28_mul(_b(a), _b(b), o, len(vals))
    assert bytes(o) == orc.fp_mul(a, b)
    for x, y in zip(vals[:30] + vals[-6:], list(reversed(vals))[:30] + vals[-5:] + [7]):
        if x == 0:
            continue
        r = _o(96)
        emul.emul_f28_lazy(_b(_to_mont(x)), _b(_to_mont(y)), r)
        assert _from_mont(bytes(r)[:48]) == (((x + y) * 8 - 2 * y) * (x - y) - x * x) % P
        assert _from_mont(bytes(r)[48:]) == pow(-x, -1, P)


def test_f28_fused_difference_of_products(emul):
    """f28_mulsub_body: a b - c d with ONE Montgomery reduction (the Y3 of the bucket loops' mixed addition), on canonical
    operands and on operands 12 x and 35 x their canonical value (the lazy magnitudes the point formulas may reach: the
    product bound is |a||b| + |c||d| < 2^11.3 p^2)."""
    vals = [int.from_bytes(os.urandom(48), "little") % P for _ in range(120)] + [0, 1, P - 1, P - 2, 1 << 380, (P - 1) // 2]
    rot = lambda k: vals[k:] + vals[:k]
    enc = lambda vs: b"".join(_to_mont(v) for v in vs)
    a, b, c, d = vals, rot(1), rot(7), rot(13)
    for scale in (1, 12, 35):
        o = _o(48 * len(vals))
        emul.emul_f28_mulsub(_b(enc(a)), _b(enc(b)), _b(enc(c)), _b(enc(d)), o, len(vals), scale)
        got = [_from_mont(bytes(o)[48 * i:48 * i + 48]) for i in range(len(vals))]
        want = [(scale * scale * (w * x - y * z)) % P for w, x, y, z in zip(a, b, c, d)]
        assert got == want, scale


def test_f28_point_formulas(emul, orc):
    """Point arithmetic on the lazy field against the oracle, with the magnitude bounds of g1_28.hpp checked."""
    n = 6
    rng = orc.rng(77)
    bases = bytearray(rng.g1_affine(n))
    scalars = bytearray(rng.fr(n))
    bases[AFF:2 * AFF] = bytes(AFF)
    scalars[2 * FR:3 * FR] = bytes(FR)
    rm1 = (0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 - 1).to_bytes(32, "little")
    scalars[3 * FR:4 * FR] = orc.fr_from_canonical_bytes(rm1)
    o = _o(AFF)
    mags = (ctypes.c_double * 3)()
    emul.emul_f28_msm(_b(bytes(bases)), _b(bytes(scalars)), n, o, mags)
    want = orc.g1_to_affine(orc.g1_msm(bytes(bases), bytes(scalars), naive=True))
    assert bytes(o) == want
    assert mags[0] <= 15.4 and mags[1] <= 15.4 and mags[2] <= 3.6, list(mags)
    # exceptional branches of the complete formulas
    g = rng.g1_affine(1)
    one = orc.fr_from_u64

    def mult(k):
        return orc.g1_to_affine(orc.g1_msm(g, one(k), naive=True))
    e = _o(8 * AFF)
    emul.emul_f28_edges(_b(g), e)
    got = [bytes(e)[i * AFF:(i + 1) * AFF] for i in range(8)]
    assert got == [mult(2), bytes(AFF), mult(4), mult(1), mult(8), bytes(AFF), mult(4), mult(3)]


def test_f28_full_xyzz_addition(emul, orc):
    """xyzz28_add / xyzz28_dbl (g1_28.hpp): the complete extended-Jacobian addition the reduction kernels add lane accumulators
    with (Y3 under one shared reduction), against the oracle: signed sums of points with ZZ != 1 as a chain and as a tree,
    repeated and opposite points among them, and the exceptional cases; magnitudes as documented."""
    n = 24
    rng = orc.rng(91)
    bases = bytearray(rng.g1_affine(n))
    bases[5 * AFF:6 * AFF] = bases[4 * AFF:5 * AFF]       # a repeated point
    bases[9 * AFF:10 * AFF] = bases[0:AFF]
    signs = bytes([(i * 7 + 3) % 3 == 0 for i in range(n)])
    signs = signs[:9] + bytes([1 - signs[0]]) + signs[10:]   # point 9 = -(point 0)
    out = _o(8 * AFF)
    mags = (ctypes.c_double * 2)()
    emul.emul_f28_xyzz_full(_b(bytes(bases)), n, _b(signs), out, mags)
    got = [bytes(out)[i * AFF:(i + 1) * AFF] for i in range(8)]
    rmod = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    sc = b"".join(orc.fr_from_canonical_bytes(((rmod - 1) if s else 1).to_bytes(32, "little")) for s in signs)
    total = orc.g1_to_affine(orc.g1_msm(bytes(bases), sc, naive=True))
    a0 = bytes(bases[:AFF])
    k0 = (rmod - 1) if signs[0] else 1

    def mult(k):
        return orc.g1_to_affine(orc.g1_msm(a0, orc.fr_from_canonical_bytes((k * k0 % rmod).to_bytes(32, "little")), naive=True))
    assert got[0] == total and got[1] == total
    assert got[2:] == [mult(2), bytes(AFF), mult(1), mult(1), mult(2), mult(3)]
    assert mags[0] <= 6.3 and mags[1] <= 2.7, list(mags)


def test_f28_sqrt_and_endomorphism_subgroup_test(emul, orc):
    """k_decompress's field work: a^((p+1)/4) and the phi(P) = [-u^2]P membership test vs the oracle's [r]P = O."""
    import random
    rnd = random.Random(5)
    on_curve = []
    while len(on_curve) < 6:
        x = rnd.randrange(P)
        rhs = (x * x * x + 4) % P
        y = pow(rhs, (P + 1) // 4, P)
        r = _o(48)
        emul.emul_f28_sqrt(_b(_to_mont(rhs)), r)
        got = _from_mont(bytes(r))
        if y * y % P == rhs:
            assert got in (y, P - y)
            on_curve.append(_to_mont(x) + _to_mont(y))
        else:
            assert got * got % P != rhs
    verdicts = [emul.emul_f28_in_subgroup(_b(pt)) for pt in on_curve]
    assert verdicts == [int(orc.g1_in_subgroup(pt)) for pt in on_curve] == [0] * 6   # random curve points have the cofactor
    good = orc.rng(3).g1_affine(4)
    for i in range(4):
        pt = good[i * AFF:(i + 1) * AFF]
        assert orc.g1_in_subgroup(pt) and emul.emul_f28_in_subgroup(_b(pt)) == 1
    assert emul.emul_f28_in_subgroup(_b(bytes(AFF))) == 1   # identity


def test_euclid_inversion_matches_fermat(emul, orc):
    """words_inv_mod_p / fe_inv_euclid / f28_inv_euclid: the one-value inversion of the batch-inversion kernels."""
    vals = [1, 2, 3, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 380, (1 << 381) - 1, 0x1234567] + [int.from_bytes(os.urandom(48), "little") % P for _ in range(40)]
    for x in vals:
        if x == 0:
            continue
        r = _o(96)
        assert emul.emul_inv_euclid(_b(_to_mont(x)), r) == 1
        assert _from_mont(bytes(r)[:48]) == pow(x, -1, P) == _from_mont(bytes(r)[48:])


def test_host_fr_product(emul, orc):
    """host_math.hpp fr64::mul (the engine's Fr algebra) vs the oracle, incl. operands near r"""
    R_ = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    edge = [orc.fr_from_canonical_bytes(v.to_bytes(32, "little")) for v in (0, 1, 2, R_ - 1, R_ - 2, (R_ - 1) // 2, 1 << 254, (1 << 254) + 12345)]
    s = orc.rng(15).fr(200) + b"".join(edge)
    t = b"".join(edge) + orc.rng(16).fr(200)
    o = _o(len(s))
    emul.emul_host_fr_mul(_b(s), _b(t), o, len(s) // 32)
    assert bytes(o) == orc.fr_mul(s, t)
    assert emul.emul_host_fr_inv_check(_b(s), len(s) // 32) == 0     # binary-Euclid inverse == Fermat, x * x^-1 == 1


def test_f28_xyzz_accumulation(emul, orc):
    """Extended-Jacobian bucket accumulation (xyzz28_add_mixed): random points, P + P, P - P, identities."""
    rng = orc.rng(91)
    pts = [rng.g1_affine(1) for _ in range(12)]
    seq = [pts[0], pts[1], pts[1], pts[2], bytes(AFF), pts[3], pts[3], pts[4]] + pts[5:] + [pts[0], pts[2]]
    signs = bytes([0, 0, 0, 1, 0, 0, 1, 0] + [i & 1 for i in range(len(pts) - 5)] + [1, 0])      # pts[3] - pts[3] cancels; pts[1] doubles
    assert len(signs) == len(seq)
    o = _o(AFF)
    mags = (ctypes.c_double * 2)()
    emul.emul_f28_xyzz_sum(_b(b"".join(seq)), len(seq), _b(signs), o, mags)
    rm1 = orc.fr_from_canonical_bytes((0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001 - 1).to_bytes(32, "little"))
    scal = b"".join(rm1 if sg else orc.fr_from_u64(1) for sg in signs)
    want = orc.g1_to_affine(orc.g1_msm(b"".join(seq), scal, naive=True))
    assert bytes(o) == want
    assert mags[0] <= 6.3 and mags[1] <= 2.7, list(mags)
    # a long bucket (the lazy, carry-pass-free differences of the inlined form alternate with the called form on every point)
    many = [rng.g1_affine(1) for _ in range(300)]
    sg = bytes([(i * 5 + i // 7) & 1 for i in range(300)])
    emul.emul_f28_xyzz_sum(_b(b"".join(many)), 300, _b(sg), o, mags)
    assert bytes(o) == orc.g1_to_affine(orc.g1_msm(b"".join(many), b"".join(rm1 if x else orc.fr_from_u64(1) for x in sg), naive=True))
    assert mags[0] <= 6.3 and mags[1] <= 2.7, list(mags)
    # a bucket that starts with the same point twice, then its negation twice: back to the identity
    seq2 = [pts[6], pts[6], pts[6], pts[6]]
    emul.emul_f28_xyzz_sum(_b(b"".join(seq2)), 4, _b(bytes([0, 0, 1, 1])), o, mags)
    assert bytes(o) == bytes(AFF)


def test_divstep_inversion_matches_euclid(emul):
    """modinv30.hpp (batched division steps, the inversion of the batch-inversion kernels) vs the binary Euclidean one."""
    import random
    rnd = random.Random(11)
    vals = [1, 2, 3, 4, 5, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 30, (1 << 30) - 1, (1 << 60) + 1, 1 << 380, (1 << 381) - 1,
            0x3fffffff << 30, P >> 1, P >> 30] + [rnd.randrange(1, P) for _ in range(3000)] + [rnd.randrange(1, 1 << b) for b in range(1, 381, 3)]
    blob = b"".join(v.to_bytes(48, "little") for v in vals)
    first = _o(48)
    assert emul.emul_modinv30_check(_b(blob), len(vals), first) == 0
    assert int.from_bytes(bytes(first), "little") == pow(vals[0], -1, P)
    for v in vals[:40]:
        o = _o(48)
        emul.emul_modinv30_check(_b(v.to_bytes(48, "little")), 1, o)
        assert int.from_bytes(bytes(o), "little") == pow(v, -1, P)


def test_fr_inversion_by_division_steps(emul):
    """fr_inv_divsteps (modinv30.hpp on 9 limbs): the inversion of the device-side protocol steps (challenge inverses) vs pow(x, -1, r)"""
    import random
    rnd = random.Random(12)
    RR = 1 << 256
    vals = [1, 2, 3, R - 1, R - 2, (R - 1) // 2, (R + 1) // 2, 1 << 30, (1 << 30) - 1, 1 << 254, (1 << 240) - 1, 0x3fffffff << 30, R >> 1, R >> 30]
    vals += [rnd.randrange(1, R) for _ in range(3000)] + [rnd.randrange(1, 1 << b) for b in range(1, 255, 2)]
    blob = b"".join((v * RR % R).to_bytes(32, "little") for v in vals)       # Montgomery wire form
    out = _o(32 * len(vals))
    emul.emul_fr_inv_divsteps(_b(blob), len(vals), out)
    out = bytes(out)
    for i, v in enumerate(vals):
        got = int.from_bytes(out[32 * i:32 * i + 32], "little")
        assert got == pow(v, -1, R) * RR % R, v
    z = _o(32)
    emul.emul_fr_inv_divsteps(_b(bytes(32)), 1, z)
    assert bytes(z) == bytes(32)                                             # 0 -> 0 (ark_ff batch_inversion convention)


def test_glv_split(emul):
    """glv.hpp: k = +-(+-|t| + q z^2) mod r with |t|, q < 2^127, and the signed radix-256 digits of both halves."""
    import random
    rnd = random.Random(5)
    z2 = 0xd201000000010000 ** 2
    vals = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, (R + 1) // 2, z2, z2 - 1, z2 + 1, z2 // 2, z2 // 2 + 1, R - z2, (R - 1) // 2 - z2 // 2,
            (1 << 255) % R, (1 << 254), (1 << 128) - 1, 1 << 127] + [rnd.randrange(R) for _ in range(4000)] + [rnd.randrange(1 << b) for b in range(1, 255)]
    blob = b"".join(v.to_bytes(32, "little") for v in vals)
    out = _o(66 * len(vals))
    emul.emul_glv_split(_b(blob), len(vals), out)
    out = bytes(out)
    for i, k in enumerate(vals):
        o = out[66 * i:66 * i + 66]
        t, q, nk, nt = int.from_bytes(o[:16], "little"), int.from_bytes(o[16:32], "little"), o[32], o[33]
        assert t < (1 << 127) - (1 << 119) and q < (1 << 127) - (1 << 119), hex(k)
        assert t <= z2 // 2 + 1
        v = (-t if nt else t) + q * z2
        assert (-v if nk else v) % R == k, hex(k)
        for half, val in ((o[34:50], t), (o[50:66], q)):
            assert sum((b - 128) << (8 * w) for w, b in enumerate(half)) == val


def test_glv_split_of_the_cofactor_has_no_upper_half(emul):
    """The endomorphism split of cpx_g1_scale's default form (k P = t P + q (-phi(P))) holds on the order-r subgroup only.  The one
    caller that scales points OUTSIDE the subgroup — the cofactor multiplication of the hash-to-curve CRS (curdleproofs_amd/crs.py,
    tests/crs.rs:38) — therefore selects the plain double-and-add (option scale_any_point).  This pins WHY the split form happened to
    give the right points there before the option existed: h = (z - 1)^2 / 3 < z^2 / 2, so q = 0 and t = h (nothing of phi is used) —
    and that scalars above z^2 / 2 do get a non-zero q, i.e. would have been wrong on such points."""
    from curdleproofs_amd.crs import COFACTOR
    z2 = 0xd201000000010000 ** 2
    assert COFACTOR < z2 // 2
    vals = [COFACTOR, z2 // 2, z2 // 2 + 2, z2, 3 * COFACTOR]
    out = _o(66 * len(vals))
    emul.emul_glv_split(_b(b"".join(v.to_bytes(32, "little") for v in vals)), len(vals), out)
    out = bytes(out)
    split = [(int.from_bytes(out[66 * i:66 * i + 16], "little"), int.from_bytes(out[66 * i + 16:66 * i + 32], "little"), out[66 * i + 32], out[66 * i + 33])
             for i in range(len(vals))]
    assert split[0] == (COFACTOR, 0, 0, 0)
    assert all(q > 0 for _, q, _, _ in split[2:])


@pytest.mark.parametrize("perwin", [0, 1])
def test_table_msm_kernel_algorithm(emul, orc, perwin):
    """The table MSM of the prover (shifted copies + endomorphism images, 128 shared buckets) and the verifier's per-window
    variant, in the kernels' arithmetic, with the two-level weighted reduction of k_reduce_sets / k_reduce_groups."""
    R_ = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    z2 = 0xd201000000010000 ** 2
    rng = orc.rng(90 + perwin)
    n = 24
    bases = bytearray(rng.g1_affine(n))
    scalars = bytearray(rng.fr(n))
    for i, v in enumerate([0, 1, R_ - 1, z2, z2 // 2, z2 // 2 + 1, (R_ - 1) // 2, (R_ + 1) // 2, 128, 129, (z2 * 128 + 128) % R_]):
        scalars[(4 + i) * FR:(5 + i) * FR] = orc.fr_from_canonical_bytes(v.to_bytes(32, "little"))
    bases[0:AFF] = bytes(AFF)                                   # identity base
    bases[3 * AFF:4 * AFF] = bases[2 * AFF:3 * AFF]             # duplicate base + same scalar: P + P in a bucket
    scalars[3 * FR:4 * FR] = scalars[2 * FR:3 * FR]
    o = _o(JAC)
    emul.emul_msm_endo(_b(bytes(bases)), _b(bytes(scalars)), n, perwin, o)
    assert orc.g1_compress_jac(bytes(o)) == orc.g1_compress_jac(orc.g1_msm(bytes(bases), bytes(scalars), naive=True))


def test_quad_reduction_programs(emul):
    """k_reduce_sets_wave (kernels.hip) runs fixed programs of (operand, operand, destination) jobs, 16 per round, over an
    array of points: quad_program.hpp.  Interpreted over integers (point = number) the plain program must return the sum of
    its 64 inputs, the bucket-set programs sum (l + 1) A_l and sum (64 + l + 1) A_l — the weights k_msm_tblw's lane
    accumulators carry (kernels.hip: lane l of set 2k + h holds bucket |d| = 64 h + l + 1) — and no job of a round may read
    what another job of the same round writes."""
    import random
    rng = random.Random(5)
    cases = [[rng.randrange(1, 1 << 40) for _ in range(64)] for _ in range(8)]
    cases += [[1 if i == k else 0 for i in range(64)] for k in range(64)]          # every single weight
    cases += [[0] * 64, [1] * 64]
    for kind, weight in ((0, lambda l: 1), (1, lambda l: l + 1), (2, lambda l: 64 + l + 1)):
        for vals in cases:
            inp = (ctypes.c_int64 * 64)(*vals)
            out = (ctypes.c_int64 * 4)()
            emul.emul_quad_program(kind, inp, out)
            assert out[0] == sum(weight(l) * v for l, v in enumerate(vals)), (kind, vals[:8])
            assert out[3] == 0, "hazard inside a round"
            assert out[1] <= 20 and out[2] == (7 if kind == 0 else 13)               # 7 / 13 addition rounds


@pytest.mark.parametrize("kind", [0, 1, 2], ids=["worker_pool", "spin_team", "spin_team_reengaged"])
def test_thread_helpers_visit_every_item_once(emul, kind):
    """host_threads.hpp: the sleeping pool of the large host loops and the spin team of a small batch's call (engaged for the
    duration of a call, helpers poll an epoch counter).  Many short jobs back to back — the pattern of the host-driven protocol,
    a loop over the proofs between every two MSM phases — must visit every item exactly once, with fewer items than threads,
    with one item, and an exception thrown by an item must reach the caller without disturbing the following jobs."""
    for threads, n, jobs in ((4, 7, 300), (8, 3, 300), (3, 1, 50), (2, 64, 100), (8, 31, 200)):
        assert emul.emul_thread_helpers(kind, threads, n, jobs, -1) == 0, (threads, n, jobs)
    assert emul.emul_thread_helpers(kind, 4, 9, 40, 17) == 0


@pytest.mark.parametrize("cb", [8, 16, 19])
def test_fixed_base_digit_recodings(emul, cb):
    """recode.hpp: the signed radix-2^cb digits k_msm_fix (fix_window_digits) and k_late_fix (FixDigitStream) index the table of
    multiples with — both forms agree, reconstruct the scalar, stay in [-2^(cb-1), 2^(cb-1) - 1], and the top window carries nothing
    out (cb = 19: 14 windows over 266 bits, chunks straddling words)."""
    import random
    rnd = random.Random(19 * cb)
    vals = [0, 1, R - 1, (R - 1) // 2, (1 << (cb - 1)) - 1, 1 << (cb - 1), (1 << cb) - 1, 1 << cb, (1 << 254) + 12345] + [rnd.randrange(R) for _ in range(200)]
    vals += [sum(((1 << (cb - 1)) - (j & 1)) << (cb * j) for j in range(255 // cb)) % R]      # every chunk at / just below the carry threshold
    blob = b"".join(v.to_bytes(32, "little") for v in vals)
    out = (ctypes.c_int32 * (64 * len(vals)))()
    W = emul.emul_fix_digits(cb, _b(blob), len(vals), out)
    assert W == -(-256 // cb)
    for i, k in enumerate(vals):
        stream, windows = list(out[64 * i:64 * i + W]), list(out[64 * i + 32:64 * i + 32 + W])
        assert stream == windows, hex(k)
        assert all(-(1 << (cb - 1)) <= d < (1 << (cb - 1)) for d in stream)
        assert sum(d << (cb * w) for w, d in enumerate(stream)) == k, hex(k)


def test_signed_nibble_recoding(emul):
    """recode.hpp recode_signed_nibbles_biased (k_late_msm): 32 digits in [-7, 8] reconstructing any value below 2^127"""
    import random
    rnd = random.Random(4)
    vals = [0, 1, 8, 9, 15, 16, (1 << 127) - 1, int("8" * 31, 16), int("9" * 31, 16), int("7" + "f" * 31, 16)] + [rnd.randrange(1 << 127) for _ in range(300)]
    out = (ctypes.c_int8 * (32 * len(vals)))()
    emul.emul_signed_nibbles(_b(b"".join(v.to_bytes(16, "little") for v in vals)), len(vals), out)
    for i, v in enumerate(vals):
        d = list(out[32 * i:32 * i + 32])
        assert all(-7 <= x <= 8 for x in d) and sum(x << (4 * j) for j, x in enumerate(d)) == v, hex(v)


@pytest.mark.parametrize("nt,split", [(2, 1), (16, 1), (16, 2), (5, 2)])
def test_late_uniform_schedule_matches_oracle_msm(emul, orc, nt, split):
    """k_late_uniform's algorithm for one output (late.hip): the shared digit sequence of the fold coefficients sorted by magnitude,
    running sums folded into totals once per magnitude, `split` lanes per output — equals the MSM  sum_t S_t P_t  of the oracle, incl.
    an identity point, equal points, zero / unit / r - 1 coefficients."""
    rng = orc.rng(1000 + nt + split)
    bases = bytearray(rng.g1_affine(nt))
    scalars = bytearray(rng.fr(nt))
    one, zero = orc.fr_from_u64(1), bytes(32)
    scalars[:32] = one
    scalars[32:64] = orc.fr_from_canonical_bytes((R - 1).to_bytes(32, "little"))
    if nt > 4:
        scalars[64:96] = zero
        bases[3 * AFF:4 * AFF] = bytes(AFF)                   # the point at infinity
        bases[4 * AFF:5 * AFF] = bases[0:AFF]                 # a repeated point
    out = _o(JAC)
    emul.emul_late_uniform(_b(bytes(bases)), _b(bytes(scalars)), nt, split, out)
    assert orc.g1_compress_jac(bytes(out)) == orc.g1_compress_jac(orc.g1_msm(bytes(bases), bytes(scalars)))


@pytest.mark.parametrize("n,slices", [(8, 1), (8, 8), (8, 4), (1, 2), (3, 8)])
def test_late_msm_straus_matches_oracle_msm(emul, orc, n, slices):
    """k_late_tables + k_late_msm + k_msm_tail for one output: multiples 1..8 and their endomorphism images, signed 4-bit digits of both
    scalar halves, the windows in `slices` groups joined by the Horner chain — equals the oracle's MSM."""
    rng = orc.rng(2000 + n + slices)
    bases, scalars = bytearray(rng.g1_affine(n)), bytearray(rng.fr(n))
    scalars[:32] = orc.fr_from_canonical_bytes((R - 1).to_bytes(32, "little"))
    if n > 2:
        scalars[32:64] = bytes(32)
        bases[2 * AFF:3 * AFF] = bases[0:AFF]
    out = _o(JAC)
    emul.emul_late_msm(_b(bytes(bases)), _b(bytes(scalars)), n, slices, out)
    assert orc.g1_compress_jac(bytes(out)) == orc.g1_compress_jac(orc.g1_msm(bytes(bases), bytes(scalars)))
