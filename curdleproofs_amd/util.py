"""Host-side mirror of the reference's `src/util.rs` public helpers on top of the C-ABI.

Same names, argument meaning and failure behaviour as the Rust functions; points and scalars are the wire
forms of `include/cpx.h` (affine 96 B, Jacobian 144 B, Fr 32 B Montgomery).  All group arithmetic runs on the
GPU through `Context`; there is no CPU path here.
"""
from . import AFF, FR, JAC
from . import params as pr

N_BLINDERS = 4   # lib.rs: N_BLINDERS


def msm(ctx, points, scalars):
    """util.rs:19-22 — panics (here: ValueError) when the lengths differ."""
    if len(points) // AFF != len(scalars) // FR or len(points) % AFF or len(scalars) % FR:
        raise ValueError("number of points != number of scalars")
    return ctx.msm(points, scalars)


def msm_from_projective(ctx, points_jac, scalars):
    """util.rs:25-29"""
    if len(points_jac) // JAC != len(scalars) // FR or len(points_jac) % JAC or len(scalars) % FR:
        raise ValueError("number of points != number of scalars")
    return ctx.msm_from_projective(points_jac, scalars)


def get_permutation(items, permutation):
    """util.rs:76-78 — `items` is a sequence (list of wire elements or of anything else)."""
    return [items[i] for i in permutation]


def inner_product(a, b):
    """util.rs:66-73 on Python integers mod r"""
    if len(a) != len(b):
        raise AssertionError("a.len() == b.len()")
    return sum(x * y for x, y in zip(a, b)) % pr.R


def get_verification_scalars_bitstring(n, logn):
    """util.rs:39-64: bitstring[i] = the rounds j whose challenge enters s_i (MSB-first bit j of i set)."""
    return [[j for j in range(logn) if (i >> (logn - j - 1)) & 1] for i in range(n)]


def shuffle_permute_and_commit_input(ctx, crs_points, ell, vec_R, vec_S, permutation, k, vec_m_blinders):
    """util.rs:83-106.  The reference draws `vec_m_blinders` from its RNG (`generate_blinders(rng, N_BLINDERS)`);
    randomness stays with the caller here (SURVEY §8b RNG contract), so the 4 blinders are an argument.

    crs_points: the ell + 7 affine points of `CurdleproofsCrs::from_points` (G[ell] | H[4] | H | G_t | G_u).
    Returns (vec_T, vec_U, M) with vec_T/vec_U affine (ell x 96 B) and M Jacobian (144 B)."""
    if len(vec_R) != ell * AFF or len(vec_S) != ell * AFF or len(permutation) != ell or len(k) != FR or len(vec_m_blinders) != N_BLINDERS * FR:
        raise ValueError("bad argument lengths")
    kR = ctx.scale(vec_R, k)                      # R.mul(k).into_affine() for every R, one launch (shared scalar)
    kS = ctx.scale(vec_S, k)
    vec_T = b"".join(get_permutation([kR[i * AFF:(i + 1) * AFF] for i in range(ell)], permutation))
    vec_U = b"".join(get_permutation([kS[i * AFF:(i + 1) * AFF] for i in range(ell)], permutation))
    sigma_ell = b"".join(pr.fr_to_wire(i) for i in permutation)          # get_permutation(range_as_fr, permutation)
    # msm(vec_G, sigma_ell) + msm(vec_H, blinders) = one MSM over G | H with the concatenated scalars
    M = msm(ctx, crs_points[:(ell + N_BLINDERS) * AFF], sigma_ell + vec_m_blinders)
    return vec_T, vec_U, M


def serialize_g1projective_vec(ctx, points_jac):
    """util.rs:126-133: compressed zcash encoding of every point, concatenated"""
    return ctx.normalize(points_jac, compressed=True)[1]


def deserialize_g1projective_vec(ctx, data, n):
    """util.rs:116-123: n compressed points -> affine wire form; raises CpxError(CPX_ERR_DESERIALIZE) on bad input"""
    if len(data) < 48 * n:
        raise ValueError("not enough bytes")
    return ctx.decompress(data[:48 * n], check_subgroup=True)
