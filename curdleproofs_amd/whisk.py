"""Byte-level Whisk API: host-side mirror of the reference's `src/whisk.rs` on top of the C-ABI (`cpx_whisk_*`).

Same function names and argument meaning as the Rust `pub fn`s; a tracker is the pair of 48-byte compressed points
`(r_G, k_r_G)` (`WhiskTracker`, whisk.rs:36-42).  Everything that touches the group runs on the GPU through `Context`;
there is no CPU path.  The reference threads an `rng` through these functions; here the RNG stays with the caller
(SURVEY 8b): each function takes the values the reference would have drawn, in the reference's order, or — when they
are omitted — draws them from the OS CSPRNG (`secrets`), which is what a production caller wants.
"""
import ctypes
import secrets

from . import AFF, FR, CPX_ERR_DESERIALIZE, CPX_OK, CpxError, _in, _out   # noqa: F401
from . import params as pr

FIELD_ELEMENT_SIZE = 32        # whisk.rs:21
G1POINT_SIZE = 48              # whisk.rs:22
TRACKER_PROOF_SIZE = 128       # whisk.rs:25
N_BLINDERS = 4


class SerializationError(CpxError):
    """ark_serialize::SerializationError of the reference's `Result`s."""

    def __init__(self, detail=""):
        super().__init__(CPX_ERR_DESERIALIZE, detail)


class WhiskTracker:
    """whisk.rs:36-42"""

    def __init__(self, r_G, k_r_G):
        if len(r_G) != G1POINT_SIZE or len(k_r_G) != G1POINT_SIZE:
            raise ValueError("a tracker is two 48-byte compressed G1 points")
        self.r_G, self.k_r_G = bytes(r_G), bytes(k_r_G)

    def to_bytes(self):
        return self.r_G + self.k_r_G

    def __eq__(self, other):
        return isinstance(other, WhiskTracker) and self.to_bytes() == other.to_bytes()

    @staticmethod
    def from_k_r(ctx, k, r):
        """whisk.rs:45-55: r_G = r * G, k_r_G = k * r_G (k, r: 32-byte wire scalars)"""
        r_G = bls_g1_scalar_multiply(ctx, g1_generator(ctx), r)
        k_r_G = bls_g1_scalar_multiply(ctx, r_G, k)
        return WhiskTracker(to_bytes_g1affine(ctx, r_G), to_bytes_g1affine(ctx, k_r_G))


def _cat(trackers):
    return b"".join(t.to_bytes() for t in trackers)


def _check(ctx, rc):
    if rc == CPX_ERR_DESERIALIZE:
        raise SerializationError(ctx._L.cpx_last_error(ctx._h).decode(errors="replace"))
    ctx._check(rc)


def _rand_fr(n):
    return pr.random_fr_wire(None, n)


def whisk_shuffle_proof_size(ctx):
    """WHISK_SHUFFLE_PROOF_SIZE (whisk.rs:23: 4496 at the reference's fixed ell = 124) for the CRS loaded into ctx"""
    return G1POINT_SIZE + ctx.proof_size


def generate_whisk_shuffle_proof(ctx, pre_trackers, permutation=None, k=None, vec_m_blinders=None, rand=None):
    """whisk.rs:144-179.  Returns (post_trackers, whisk_shuffle_proof_bytes).  ctx must hold the CRS (`Context.set_crs`).
    permutation / k / vec_m_blinders / rand are the reference's rng draws (shuffle of 0..ell, Fr::rand, 4 blinders, the
    3n+9 draws of CurdleproofsProof::new); omitted values come from the OS CSPRNG."""
    ell, n = ctx.ell, ctx.n
    if len(pre_trackers) != ell:
        raise ValueError("need exactly ell = %d trackers" % ell)
    if permutation is None:
        permutation = list(range(ell))
        for i in range(ell - 1, 0, -1):                      # Fisher-Yates on the CSPRNG
            j = secrets.randbelow(i + 1)
            permutation[i], permutation[j] = permutation[j], permutation[i]
    k = _rand_fr(1) if k is None else k
    vec_m_blinders = _rand_fr(N_BLINDERS) if vec_m_blinders is None else vec_m_blinders
    rand = _rand_fr(3 * n + 9) if rand is None else rand
    if len(permutation) != ell or len(k) != FR or len(vec_m_blinders) != N_BLINDERS * FR or len(rand) != (3 * n + 9) * FR:
        raise ValueError("bad argument lengths")
    post, proof = _out(ell * 2 * G1POINT_SIZE), _out(whisk_shuffle_proof_size(ctx))
    perm = (ctypes.c_uint32 * ell)(*permutation)
    _check(ctx, ctx._L.cpx_whisk_generate_shuffle_proof(ctx._h, _in(_cat(pre_trackers)), perm, _in(k), _in(vec_m_blinders), _in(rand), post, proof))
    pb = bytes(post)
    return [WhiskTracker(pb[96 * i:96 * i + 48], pb[96 * i + 48:96 * i + 96]) for i in range(ell)], bytes(proof)


def is_valid_whisk_shuffle_proof(ctx, pre_trackers, post_trackers, whisk_shuffle_proof_bytes, rand=None):
    """whisk.rs:106-130: True / False, or SerializationError for undecodable trackers / proof bytes.
    rand: the verifier's eight accumulate_check factors (msm_accumulator.rs:44); drawn from the CSPRNG when omitted."""
    ell = ctx.ell
    if len(pre_trackers) != ell or len(post_trackers) != ell:
        raise ValueError("need exactly ell = %d trackers on both sides" % ell)
    if len(whisk_shuffle_proof_bytes) != whisk_shuffle_proof_size(ctx):
        raise SerializationError("wrong proof length")      # a fixed-size array in the reference
    rand = _rand_fr(8) if rand is None else rand
    if len(rand) != 8 * FR:
        raise ValueError("8 random factors")
    valid = ctypes.c_int(0)
    _check(ctx, ctx._L.cpx_whisk_is_valid_shuffle_proof(ctx._h, _in(_cat(pre_trackers)), _in(_cat(post_trackers)), _in(whisk_shuffle_proof_bytes), _in(rand),
                                                        ctypes.byref(valid)))
    return bool(valid.value)


def generate_whisk_tracker_proof(ctx, tracker, k, blinder=None):
    """whisk.rs:228-263: 128-byte proof of knowledge of k with tracker.k_r_G == k * tracker.r_G"""
    blinder = _rand_fr(1) if blinder is None else blinder
    if len(k) != FR or len(blinder) != FR:
        raise ValueError("k and blinder are 32-byte wire scalars")
    out = _out(TRACKER_PROOF_SIZE)
    _check(ctx, ctx._L.cpx_whisk_generate_tracker_proof(ctx._h, _in(tracker.to_bytes()), _in(k), _in(blinder), out))
    return bytes(out)


def is_valid_whisk_tracker_proof(ctx, tracker, k_commitment, tracker_proof):
    """whisk.rs:183-226"""
    if len(k_commitment) != G1POINT_SIZE or len(tracker_proof) != TRACKER_PROOF_SIZE:
        raise SerializationError("wrong length")
    valid = ctypes.c_int(0)
    _check(ctx, ctx._L.cpx_whisk_is_valid_tracker_proof(ctx._h, _in(tracker.to_bytes()), _in(k_commitment), _in(tracker_proof), ctypes.byref(valid)))
    return bool(valid.value)


# ---- small helpers of whisk.rs:295-345 ----
def to_bytes_g1affine(ctx, g1):
    """whisk.rs:307-311: affine wire point (96 B) -> 48-byte compressed encoding"""
    from . import JAC
    one = pr.fp_to_wire(1)
    jac = g1 + (bytes(48) if g1 == bytes(AFF) else one)
    assert len(jac) == JAC
    return ctx.normalize(jac, compressed=True)[1]


def from_bytes_g1affine(ctx, buf):
    """whisk.rs:313-315: validated decompression (on curve, in the subgroup)"""
    try:
        return ctx.decompress(buf, check_subgroup=True)
    except CpxError as e:
        if e.code == CPX_ERR_DESERIALIZE:
            raise SerializationError(str(e))
        raise


def g1_generator(ctx):
    """whisk.rs:318-320"""
    return pr.g1_generator_wire()


def bls_g1_scalar_multiply(ctx, g1, scalar):
    """whisk.rs:323-325"""
    return ctx.scale(g1, scalar)


def rand_scalar():
    """whisk.rs:328-330 (CSPRNG instead of a caller-supplied rng)"""
    return _rand_fr(1)


def to_bytes_fr(fr):
    """whisk.rs:333-337: wire scalar -> 32-byte little-endian canonical"""
    return pr.fr_from_wire(fr).to_bytes(32, "little")


def from_bytes_fr(b):
    """whisk.rs:343-345: little-endian bytes, reduced mod r"""
    return pr.fr_to_wire(int.from_bytes(b, "little"))
