"""ctypes front-end to the CPU oracle (oracle/_build/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by bench.py's cpu_baseline leg and by
__graft_entry__.smoke() as the checker.  Nothing under curdleproofs_amd/ imports this module.

Wire layouts (same as include/cpx.h): Fr = 32 B (4 x u64 LE Montgomery), affine point = 96 B
(x || y Montgomery limbs, identity = zeros), Jacobian point = 144 B.
"""
import ctypes
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB = os.path.join(_ORACLE_DIR, "_build", "liborc.so")

N_BLINDERS = 4
FR = 32
AFF = 96
JAC = 144


_LIB_FAST = os.path.join(_ORACLE_DIR, "_build", "liborc_fast.so")


def fast_build_runs_here():
    """liborc_fast.so is compiled with -march=x86-64-v3 -madx (mulx / adcx / adox): only loadable where the CPU has BMI2 and ADX"""
    try:
        with open("/proc/cpuinfo") as f:
            flags = next(l for l in f if l.startswith("flags")).split()
        return "bmi2" in flags and "adx" in flags and "avx2" in flags
    except Exception:
        return False


def oracle_build_flags(fast=False):
    """the compiler flags of the (fast) oracle build, as oracle/Makefile states them"""
    with open(os.path.join(_ORACLE_DIR, "Makefile")) as f:
        for l in f:
            if l.startswith("FASTFLAGS ?=" if fast else "CXXFLAGS ?="):
                return "g++ " + l.split("?=", 1)[1].strip()
    return None


def build_oracle(force=False, fast=False):
    lib = _LIB_FAST if fast else _LIB
    srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR) if f.endswith((".h", ".cpp")) or f == "Makefile"]
    stale = (not os.path.exists(lib)) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s", "_build/" + os.path.basename(lib)] + (["-B"] if force else []))
    return lib


def _buf(n):
    return (ctypes.c_uint8 * n)()


def _cb(b):
    return (ctypes.c_uint8 * len(b)).from_buffer_copy(b)


class Oracle:
    def __init__(self, fast=False):
        """fast: the ORC_FAST build (ark-ff's unrolled no-carry CIOS on mulx + dedicated squaring, -march=x86-64-v3 -madx) — the CPU
        baseline of bench.py; the tests check it against the same known-answer vectors as the portable build"""
        self.fast = bool(fast)
        self.lib = ctypes.CDLL(build_oracle(fast=fast))
        L = self.lib
        L.orc_rng_new.restype = ctypes.c_void_p
        L.orc_rng_new.argtypes = [ctypes.c_uint64]
        L.orc_rng_u32.restype = ctypes.c_uint32
        vp_, sz_ = ctypes.c_void_p, ctypes.c_size_t
        L.orc_rng_free.argtypes = [vp_]
        L.orc_rng_u32.argtypes = [vp_]
        L.orc_rng_key.argtypes = [vp_, vp_]
        L.orc_rng_fr.argtypes = [vp_, vp_, sz_]
        L.orc_rng_g1_affine.argtypes = [vp_, vp_, sz_]
        L.orc_rng_shuffle.argtypes = [vp_, vp_, sz_]
        sz = ctypes.c_size_t
        vp = ctypes.c_void_p
        L.orc_challenges.argtypes = [vp, sz, vp]
        L.orc_kat_shuffle_proof.argtypes = [sz, vp, vp, vp, sz, ctypes.POINTER(sz)]
        L.orc_whisk_tracker_proof.argtypes = [vp, vp, vp, vp]
        L.orc_whisk_tracker_verify.argtypes = [vp, vp, vp]
        L.orc_whisk_shuffle_verify.argtypes = [sz, vp, vp, vp, vp, sz, vp, sz]
        L.orc_fp_mul.argtypes = [vp, vp, vp, sz]
        L.orc_fp_inv.argtypes = [vp, vp, sz]
        L.orc_fr_mul.argtypes = [vp, vp, vp, sz]
        L.orc_fr_inv.argtypes = [vp, vp, sz]
        L.orc_fr_from_u64.argtypes = [ctypes.c_uint64, vp]
        L.orc_fr_to_canonical_bytes.argtypes = [vp, vp, sz]
        L.orc_fr_from_canonical_bytes.argtypes = [vp, vp, sz]
        L.orc_fr_from_le_bytes_mod_order.argtypes = [vp, sz, vp]
        L.orc_g1_msm.argtypes = [vp, vp, sz, vp]
        L.orc_g1_msm_naive.argtypes = [vp, vp, sz, vp]
        L.orc_g1_msm_jac.argtypes = [vp, vp, sz, vp]
        L.orc_g1_fold.argtypes = [vp, vp, vp, sz]
        L.orc_g1_scale.argtypes = [vp, vp, sz, sz, vp]
        L.orc_g1_to_affine.argtypes = [vp, sz, vp]
        L.orc_g1_add_jac.argtypes = [vp, vp, vp]
        L.orc_g1_eq_jac.argtypes = [vp, vp]
        L.orc_g1_compress.argtypes = [vp, sz, vp]
        L.orc_g1_compress_jac.argtypes = [vp, sz, vp]
        L.orc_g1_decompress.argtypes = [vp, sz, vp]
        L.orc_generate_crs_points.argtypes = [sz, vp]
        L.orc_crs_sums.argtypes = [sz, vp, vp, vp]
        L.orc_make_instance.argtypes = [sz, ctypes.c_uint64] + [vp] * 10 + [vp, sz, ctypes.POINTER(sz), vp]
        L.orc_shuffle_permute_and_commit_input.argtypes = [sz] + [vp] * 9
        L.orc_prove.argtypes = [sz] + [vp] * 10 + [sz, vp, sz, ctypes.POINTER(sz)]
        L.orc_verify.argtypes = [sz] + [vp] * 7 + [sz, vp, sz]
        L.orc_time_prove_verify.argtypes = [sz] + [vp] * 10 + [sz, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                                                ctypes.POINTER(ctypes.c_double)]

    def set_strict_infinity(self, strict):
        """how encodings with the infinity flag are read (oracle/g1.h): False = ark-bls12-381 ^0.4 (the identity whatever the other bits say),
        True = only 0xc0 || 0^47.  Process-wide for this library."""
        self.lib.orc_set_strict_infinity(1 if strict else 0)

    # ---- self checks ----
    def merlin_test_vector(self):
        o = _buf(32)
        self.lib.orc_merlin_test_vector(o)
        return bytes(o)

    def keccak_f1600(self, state200):
        b = _cb(state200)
        self.lib.orc_keccak_f1600(b)
        return bytes(b)

    def challenges(self, msg):
        o = _buf(64)
        self.lib.orc_challenges(_cb(msg), len(msg), o)
        return bytes(o)

    def kat_tracker_proof(self):
        o = _buf(128)
        v = self.lib.orc_kat_tracker_proof(o)
        return v, bytes(o)

    def generate_whisk_tracker_proof(self, tracker96, k, blinder):
        o = _buf(128)
        if self.lib.orc_whisk_tracker_proof(_cb(tracker96), _cb(k), _cb(blinder), o) != 1:
            raise ValueError("undecodable tracker")
        return bytes(o)

    def is_valid_whisk_tracker_proof(self, tracker96, k_commitment, proof):
        """1 valid, 0 invalid, -1 deserialisation error"""
        return self.lib.orc_whisk_tracker_verify(_cb(tracker96), _cb(k_commitment), _cb(proof))

    def is_valid_whisk_shuffle_proof(self, ell, crs_points, pre, post, proof, rand):
        return self.lib.orc_whisk_shuffle_verify(ell, _cb(crs_points), _cb(pre), _cb(post), _cb(proof), len(proof), _cb(rand), len(rand) // FR)

    def kat_shuffle_proof(self, ell=124):
        pre, post, proof = _buf(96 * ell), _buf(96 * ell), _buf(8192)
        n = ctypes.c_size_t(0)
        v = self.lib.orc_kat_shuffle_proof(ell, pre, post, proof, 8192, ctypes.byref(n))
        return v, bytes(proof[: n.value]), bytes(pre), bytes(post)

    # ---- RNG model ----
    class Rng:
        def __init__(self, orc, seed):
            self.o = orc
            self.h = ctypes.c_void_p(orc.lib.orc_rng_new(seed))

        def __del__(self):
            try:
                self.o.lib.orc_rng_free(self.h)
            except Exception:
                pass

        def u32(self):
            return self.o.lib.orc_rng_u32(self.h)

        def key(self):
            b = _buf(32)
            self.o.lib.orc_rng_key(self.h, b)
            return bytes(b)

        def fr(self, n=1):
            b = _buf(FR * n)
            self.o.lib.orc_rng_fr(self.h, b, n)
            return bytes(b)

        def g1_affine(self, n=1):
            b = _buf(AFF * n)
            self.o.lib.orc_rng_g1_affine(self.h, b, n)
            return bytes(b)

        def shuffle(self, n):
            p = (ctypes.c_uint32 * n)(*range(n))
            self.o.lib.orc_rng_shuffle(self.h, p, n)
            return list(p)

    def rng(self, seed):
        return Oracle.Rng(self, seed)

    # ---- primitives ----
    def fp_mul(self, a, b):
        n = len(a) // 48
        o = _buf(48 * n)
        self.lib.orc_fp_mul(_cb(a), _cb(b), o, n)
        return bytes(o)

    def fp_inv(self, a):
        n = len(a) // 48
        o = _buf(48 * n)
        self.lib.orc_fp_inv(_cb(a), o, n)
        return bytes(o)

    def fr_mul(self, a, b):
        n = len(a) // FR
        o = _buf(FR * n)
        self.lib.orc_fr_mul(_cb(a), _cb(b), o, n)
        return bytes(o)

    def fr_inv(self, a):
        n = len(a) // FR
        o = _buf(FR * n)
        self.lib.orc_fr_inv(_cb(a), o, n)
        return bytes(o)

    def fr_from_u64(self, x):
        o = _buf(FR)
        self.lib.orc_fr_from_u64(x, o)
        return bytes(o)

    def fr_to_canonical_bytes(self, a):
        n = len(a) // FR
        o = _buf(FR * n)
        self.lib.orc_fr_to_canonical_bytes(_cb(a), o, n)
        return bytes(o)

    def fr_from_canonical_bytes(self, a):
        n = len(a) // FR
        o = _buf(FR * n)
        if self.lib.orc_fr_from_canonical_bytes(_cb(a), o, n) != 0:
            raise ValueError("not a canonical Fr encoding")
        return bytes(o)

    def fr_from_le_bytes_mod_order(self, b):
        o = _buf(FR)
        self.lib.orc_fr_from_le_bytes_mod_order(_cb(b), len(b), o)
        return bytes(o)

    def g1_generator(self):
        o = _buf(AFF)
        self.lib.orc_g1_generator(o)
        return bytes(o)

    def g1_on_curve(self, p):
        return bool(self.lib.orc_g1_on_curve(_cb(p)))

    def g1_in_subgroup(self, p):
        return bool(self.lib.orc_g1_in_subgroup(_cb(p)))

    def g1_msm(self, bases, scalars, naive=False):
        n = len(scalars) // FR
        assert len(bases) == AFF * n
        o = _buf(JAC)
        (self.lib.orc_g1_msm_naive if naive else self.lib.orc_g1_msm)(_cb(bases) if n else None, _cb(scalars) if n else None, n, o)
        return bytes(o)

    def g1_msm_jac(self, bases, scalars):
        n = len(scalars) // FR
        o = _buf(JAC)
        self.lib.orc_g1_msm_jac(_cb(bases), _cb(scalars), n, o)
        return bytes(o)

    def g1_fold(self, PL, PR, gamma):
        half = len(PL) // AFF
        b = _cb(PL)
        self.lib.orc_g1_fold(b, _cb(PR), _cb(gamma), half)
        return bytes(b)

    def g1_scale(self, P, scalars):
        n = len(P) // AFF
        stride = 0 if len(scalars) == FR else FR   # one scalar = broadcast
        o = _buf(AFF * n)
        self.lib.orc_g1_scale(_cb(P), _cb(scalars), stride, n, o)
        return bytes(o)

    def g1_to_affine(self, jac):
        n = len(jac) // JAC
        o = _buf(AFF * n)
        self.lib.orc_g1_to_affine(_cb(jac), n, o)
        return bytes(o)

    def g1_add_jac(self, a, b):
        o = _buf(JAC)
        self.lib.orc_g1_add_jac(_cb(a), _cb(b), o)
        return bytes(o)

    def g1_eq_jac(self, a, b):
        return bool(self.lib.orc_g1_eq_jac(_cb(a), _cb(b)))

    def g1_compress(self, aff):
        n = len(aff) // AFF
        o = _buf(48 * n)
        self.lib.orc_g1_compress(_cb(aff), n, o)
        return bytes(o)

    def g1_compress_jac(self, jac):
        n = len(jac) // JAC
        o = _buf(48 * n)
        self.lib.orc_g1_compress_jac(_cb(jac), n, o)
        return bytes(o)

    def g1_decompress(self, comp):
        n = len(comp) // 48
        o = _buf(AFF * n)
        if self.lib.orc_g1_decompress(_cb(comp), n, o) != 0:
            raise ValueError("invalid compressed G1 point")
        return bytes(o)

    # ---- protocol ----
    def generate_crs_points(self, ell):
        cache = self.__dict__.setdefault("_crs_cache", {})
        if ell not in cache:
            o = _buf(AFF * (ell + 7))
            self.lib.orc_generate_crs_points(ell, o)
            cache[ell] = bytes(o)
        return cache[ell]

    def hash_to_curve_points(self, num_points, seed):
        """/root/reference/tests/crs.rs:13-52 generate_random_points: (points as 96-byte affine records, increments consumed)"""
        o = _buf(AFF * num_points)
        self.lib.orc_hash_to_curve_points.restype = ctypes.c_uint64
        s = seed.encode() if isinstance(seed, str) else bytes(seed)
        inc = self.lib.orc_hash_to_curve_points(_cb(s), ctypes.c_size_t(len(s)), ctypes.c_size_t(num_points), o)
        return bytes(o), inc

    def sha256(self, msg):
        o = _buf(32)
        self.lib.orc_sha256(_cb(msg), ctypes.c_size_t(len(msg)), o)
        return bytes(o)

    def crs_sums(self, ell, crs_points):
        g, h = _buf(AFF), _buf(AFF)
        self.lib.orc_crs_sums(ell, _cb(crs_points), g, h)
        return bytes(g), bytes(h)

    def make_instance(self, ell, seed, crs_points):
        """README.md:76-118 recipe for StdRng(seed): instance, witness, the prover's draws, the oracle's proof and verdict.  Results are
        memoised (the engine-variant tests ask for the same few instances a dozen times; an ell = 252 instance costs 0.7 s)."""
        key = (ell, seed, hash(crs_points))
        cache = self.__dict__.setdefault("_inst_cache", {})
        if key in cache:
            return dict(cache[key])
        inst = self._make_instance(ell, seed, crs_points)
        if len(cache) < 256:
            cache[key] = dict(inst)
        return inst

    def _make_instance(self, ell, seed, crs_points):
        n = ell + N_BLINDERS
        R, S, T, U = (_buf(AFF * ell) for _ in range(4))
        M, k, mb = _buf(JAC), _buf(FR), _buf(FR * 4)
        perm = (ctypes.c_uint32 * ell)()
        prand, vrand = _buf(FR * (3 * n + 9)), _buf(FR * 8)
        proof, plen = _buf(16384), ctypes.c_size_t(0)
        v = self.lib.orc_make_instance(ell, seed, _cb(crs_points), R, S, T, U, M, perm, k, mb, prand, proof, 16384,
                                       ctypes.byref(plen), vrand)
        if v < 0:
            raise RuntimeError("orc_make_instance failed: %d" % v)
        return dict(ell=ell, seed=seed, vec_R=bytes(R), vec_S=bytes(S), vec_T=bytes(T), vec_U=bytes(U), M=bytes(M),
                    permutation=list(perm), k=bytes(k), vec_m_blinders=bytes(mb), prover_rand=bytes(prand),
                    proof=bytes(proof[: plen.value]), verifier_rand=bytes(vrand), verdict=v)

    def shuffle_permute_and_commit_input(self, ell, crs_points, vec_R, vec_S, permutation, k, m_blinders):
        T, U, M = _buf(AFF * ell), _buf(AFF * ell), _buf(JAC)
        perm = (ctypes.c_uint32 * ell)(*permutation)
        rc = self.lib.orc_shuffle_permute_and_commit_input(ell, _cb(crs_points), _cb(vec_R), _cb(vec_S), perm, _cb(k),
                                                           _cb(m_blinders), T, U, M)
        assert rc == 0
        return bytes(T), bytes(U), bytes(M)

    def prove(self, ell, crs_points, vec_R, vec_S, vec_T, vec_U, M, permutation, k, m_blinders, rand):
        perm = (ctypes.c_uint32 * ell)(*permutation)
        proof, plen = _buf(16384), ctypes.c_size_t(0)
        rc = self.lib.orc_prove(ell, _cb(crs_points), _cb(vec_R), _cb(vec_S), _cb(vec_T), _cb(vec_U), _cb(M), perm, _cb(k),
                                _cb(m_blinders), _cb(rand), len(rand) // FR, proof, 16384, ctypes.byref(plen))
        if rc != 0:
            raise RuntimeError("orc_prove failed: %d" % rc)
        return bytes(proof[: plen.value])

    def verify(self, ell, crs_points, vec_R, vec_S, vec_T, vec_U, M, proof, rand):
        return self.lib.orc_verify(ell, _cb(crs_points), _cb(vec_R), _cb(vec_S), _cb(vec_T), _cb(vec_U), _cb(M), _cb(proof),
                                   len(proof), _cb(rand), len(rand) // FR)

    def time_prove_verify(self, inst, crs_points, reps=1):
        ell = inst["ell"]
        perm = (ctypes.c_uint32 * ell)(*inst["permutation"])
        tp, tv = ctypes.c_double(0), ctypes.c_double(0)
        ok = self.lib.orc_time_prove_verify(ell, _cb(crs_points), _cb(inst["vec_R"]), _cb(inst["vec_S"]), _cb(inst["vec_T"]),
                                            _cb(inst["vec_U"]), _cb(inst["M"]), perm, _cb(inst["k"]), _cb(inst["vec_m_blinders"]),
                                            _cb(inst["prover_rand"]), len(inst["prover_rand"]) // FR, _cb(inst["verifier_rand"]),
                                            reps, ctypes.byref(tp), ctypes.byref(tv))
        return ok, tp.value, tv.value
