"""CRS hex/JSON tooling: host-side mirror of the reference's `CurdleproofsCrsHex` (src/crs.rs:77-142).

The JSON object has the fields vec_G, vec_H, H, G_t, G_u, G_sum, H_sum; every point is "0x" + the hex of its 48-byte
compressed encoding.  Decoding validates every point on the GPU (on curve, in the subgroup: `from_bytes_g1affine`); G_sum /
H_sum are recomputed by `Context.set_crs` and compared with the file's, so a CRS whose sums do not match its vectors is
refused (the reference trusts the file).  No CPU group arithmetic here.
"""
import json
import re

from . import AFF, N_BLINDERS, CPX_ERR_DESERIALIZE, CpxError
from . import whisk


_HEX_POINT = re.compile(r"0x[0-9a-fA-F]{96}")   # used with fullmatch: `$` alone would let a trailing newline through


def crs_to_hex(ctx):
    """`CurdleproofsCrsHex::try_from(&crs)` for the CRS loaded into ctx: dict of hex strings"""
    ell, n = ctx.ell, ctx.n
    pts = ctx.crs_points
    g_sum, h_sum = ctx.crs_sums()
    allp = pts + g_sum + h_sum
    jac = b"".join(allp[i * AFF:(i + 1) * AFF] + (bytes(48) if allp[i * AFF:(i + 1) * AFF] == bytes(AFF) else whisk.pr.fp_to_wire(1)) for i in range(ell + 9))
    comp = ctx.normalize(jac, compressed=True)[1]
    hx = ["0x" + comp[48 * i:48 * i + 48].hex() for i in range(ell + 9)]
    return dict(vec_G=hx[:ell], vec_H=hx[ell:n], H=hx[n], G_t=hx[n + 1], G_u=hx[n + 2], G_sum=hx[n + 3], H_sum=hx[n + 4])


def crs_from_hex(ctx, obj):
    """`TryInto<CurdleproofsCrs> for &CurdleproofsCrsHex` followed by `Context.set_crs`: loads the CRS into ctx.
    Raises whisk.SerializationError for a malformed string / point (SerializationError::InvalidData in the reference)."""
    def unhex(s):
        if not isinstance(s, str) or not s.startswith("0x"):
            raise whisk.SerializationError("missing 0x prefix")
        if not _HEX_POINT.fullmatch(s):   # hex::decode rejects whitespace and odd lengths; bytes.fromhex alone would skip blanks
            raise whisk.SerializationError("a compressed G1 point is 0x + 96 hex digits")
        return bytes.fromhex(s[2:])
    vec_G, vec_H = [unhex(s) for s in obj["vec_G"]], [unhex(s) for s in obj["vec_H"]]
    if len(vec_H) != N_BLINDERS:
        raise ValueError("vec_H must hold N_BLINDERS = 4 points")
    ell = len(vec_G)
    comp = b"".join(vec_G + vec_H + [unhex(obj["H"]), unhex(obj["G_t"]), unhex(obj["G_u"]), unhex(obj["G_sum"]), unhex(obj["H_sum"])])
    try:
        pts = ctx.decompress(comp, check_subgroup=True)
    except CpxError as e:
        if e.code == CPX_ERR_DESERIALIZE:
            raise whisk.SerializationError(str(e))
        raise
    # the sums are checked BEFORE the CRS is loaded (tier-0 point sums on the device): a refused file leaves ctx as it was
    one = whisk.pr.fp_to_wire(1)
    jac = lambda lo, hi: b"".join(pts[i * AFF:(i + 1) * AFF] + (bytes(48) if pts[i * AFF:(i + 1) * AFF] == bytes(AFF) else one) for i in range(lo, hi))
    sums = ctx.normalize(ctx.sum_jac(jac(0, ell))[0] + ctx.sum_jac(jac(ell, ell + N_BLINDERS))[0])
    if sums != pts[(ell + 7) * AFF:(ell + 9) * AFF]:
        raise ValueError("G_sum / H_sum of the file are not the sums of vec_G / vec_H")
    ctx.set_crs(ell, pts[:(ell + 7) * AFF])
    return ell


def crs_to_json(ctx):
    return json.dumps(crs_to_hex(ctx))


def crs_from_json(ctx, text):
    return crs_from_hex(ctx, json.loads(text))


# ---- hash-to-curve CRS: /root/reference/tests/crs.rs:13-52 `generate_random_points` -------------------------------------
COFACTOR = 0x396c8c005555e1568c00aaab0000aaab   # |E(Fp)| / r: `mul_by_cofactor` multiplies by the full cofactor (not h_eff)
CRS_EXTRA_POINTS = 3                             # crs.rs: H, G_t, G_u


def generate_random_points(ctx, num_points, seed):
    """Try-and-increment hash to the curve, as the reference's CRS ceremony test does it: for increment = 0, 1, ...
    hash = SHA-256(seed || be64(increment)); x = the integer whose 64-bit limbs 0..3 are the big-endian words of the hash
    (`BigInteger384::new([u64; 6])`, limbs 4 and 5 zero); if x^3 + 4 is a square keep `get_point_from_x_unchecked(x, false)` (the
    smaller y) times the cofactor, provided it lies in the subgroup.  Returns (num_points affine points of 96 bytes, increments
    consumed).  The hashing is host work (hashlib); square roots, cofactor multiplication and the subgroup test run on the
    GPU in batches of candidates through the C-ABI: cpx_g1_decompress_status, cpx_g1_scale, cpx_g1_normalize."""
    seed = seed.encode() if isinstance(seed, str) else bytes(seed)
    h_wire = whisk.pr.fr_to_wire(COFACTOR)          # the cofactor as a scalar (< r)
    # The candidates are points of E(Fp), not of the order-r subgroup: cpx_g1_scale's default form splits the scalar by the G1 endomorphism,
    # an identity of the subgroup only.  Option scale_any_point selects the plain 257-step double-and-add, which holds on all of E(Fp).
    ctx.set_option("scale_any_point", 1)
    try:
        return _generate_random_points(ctx, num_points, seed, h_wire)
    finally:
        ctx.set_option("scale_any_point", 0)


def _generate_random_points(ctx, num_points, seed, h_wire):
    import hashlib
    points, increment = [], 0
    while len(points) < num_points:
        batch = max(16, 2 * (num_points - len(points)) + 8)      # about half of the candidates have a square root
        cands = []
        for inc in range(increment, increment + batch):
            hsh = hashlib.sha256(seed + inc.to_bytes(8, "big")).digest()
            x = sum(int.from_bytes(hsh[8 * i:8 * i + 8], "big") << (64 * i) for i in range(4))
            cands.append(bytes([0x80 | (x >> 376)]) + (x & ((1 << 376) - 1)).to_bytes(47, "big"))   # compressed, sort flag clear = smaller y
        aff, status = ctx.decompress_status(b"".join(cands), check_subgroup=False)
        hits = [j for j in range(batch) if status[j] == 0]
        scaled = ctx.scale(b"".join(aff[AFF * j:AFF * (j + 1)] for j in hits), h_wire) if hits else b""
        one = whisk.pr.fp_to_wire(1)
        jac = b"".join(scaled[AFF * t:AFF * (t + 1)] + (bytes(48) if scaled[AFF * t:AFF * (t + 1)] == bytes(AFF) else one) for t in range(len(hits)))
        comp = ctx.normalize(jac, compressed=True)[1] if hits else b""
        in_group = ctx.decompress_status(comp, check_subgroup=True)[1] if hits else []   # is_in_correct_subgroup_assuming_on_curve
        for t, j in enumerate(hits):
            if in_group[t] == 0:
                points.append(scaled[AFF * t:AFF * (t + 1)])
                if len(points) == num_points:
                    return b"".join(points), increment + j + 1
        increment += batch
    return b"".join(points), increment


def crs_from_seed(ctx, ell, seed):
    """tests/crs.rs:54-66: `CurdleproofsCrs::from_points(ell, generate_random_points(n + CRS_EXTRA_POINTS, seed))` loaded into ctx;
    returns the number of hash attempts"""
    pts, attempts = generate_random_points(ctx, ell + N_BLINDERS + CRS_EXTRA_POINTS, seed)
    ctx.set_crs(ell, pts)
    return attempts
