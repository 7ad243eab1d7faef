"""curdleproofs_amd — MI355X (gfx950) core for the Curdleproofs shuffle argument's G1 hot path.

Python mirror of the reference's interface for this path (asn-d6/curdleproofs):

    util::msm / msm_from_projective           -> Context.msm / Context.msm_from_projective
    the fold / rescale loops                  -> Context.fold / Context.scale
    msm_accumulator::MsmAccumulator           -> MsmAccumulator
    CurdleproofsCrs::from_points              -> Context.set_crs
    CurdleproofsProof::{new, verify}          -> Context.prove_batch / Context.verify_batch
                                                 (CurdleproofsProof.new / .verify for one instance)

All of it is a thin ctypes layer over the C-ABI in include/cpx.h (curdleproofs_amd/_lib/libcpx.so).
There is no CPU implementation behind these calls: if the HIP library or a GPU is missing they raise.
Byte layouts are arkworks' in-memory limbs (Fr 32 B, affine 96 B, Jacobian 144 B); see include/cpx.h.
"""
import ctypes
import os

# eight hardware queues for the HIP streams of this process (the runtime's default of four lets two streams of one engine context share a
# queue in about half of the runs of a two-context process: capi.cpp); read by the runtime at its first call, so it is set
# as early as this package is imported — libcpx.so sets it again when it is loaded — and a value the caller exported wins
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .build import LIB as _LIB_PATH

FR = 32
AFF = 96
JAC = 144
N_BLINDERS = 4   # reference src/lib.rs:35

CPX_OK = 0
CPX_ERR_ARG = -1
CPX_ERR_NOT_POW2 = -2
CPX_ERR_HIP = -3
CPX_ERR_VERIFY = -4
CPX_ERR_DESERIALIZE = -5
CPX_ERR_STATE = -6
CPX_ERR_INTERNAL = -7

_ERRNAMES = {CPX_ERR_ARG: "bad argument", CPX_ERR_NOT_POW2: "ell + 4 is not a power of two", CPX_ERR_HIP: "HIP failure",
             CPX_ERR_VERIFY: "verification failed", CPX_ERR_DESERIALIZE: "deserialization failed",
             CPX_ERR_STATE: "call order (CRS / batch not set)", CPX_ERR_INTERNAL: "internal error"}


class CpxError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__("cpx error %d (%s) %s" % (code, _ERRNAMES.get(code, "?"), detail))


class ProofError(CpxError):
    """Mirror of the reference's ProofError::VerificationError (src/errors.rs:9)."""


_lib = None
_libs = {}   # realpath -> loaded library (the default one and build variants for A/B runs)

EXPORTS = [
    "cpx_host_alloc", "cpx_host_free", "cpx_ctx_create", "cpx_ctx_destroy", "cpx_last_error", "cpx_device_count", "cpx_ctx_set_option", "cpx_ctx_get_option", "cpx_ctx_set_crs", "cpx_crs_sums", "cpx_proof_size", "cpx_batch_size",
    "cpx_g1_msm", "cpx_g1_msm_jac", "cpx_g1_fold", "cpx_g1_scale", "cpx_g1_normalize", "cpx_g1_decompress", "cpx_g1_decompress_status",
    "cpx_accum_new", "cpx_accum_free", "cpx_accum_check", "cpx_accum_verify",
    "cpx_batch_load", "cpx_batch_load_begin", "cpx_batch_load_end", "cpx_batch_prove", "cpx_batch_verify", "cpx_batch_verify_fused", "cpx_g1_sum_jac",
    "cpx_whisk_generate_shuffle_proof", "cpx_whisk_is_valid_shuffle_proof", "cpx_whisk_generate_tracker_proof", "cpx_whisk_is_valid_tracker_proof",
    "cpx_set_profiling", "cpx_reset_stats", "cpx_get_stat", "cpx_set_host_threads", "cpx_bench_fpmul",
]


def load_library(path=None):
    """Loads libcpx.so; raises if it has not been built (python -m curdleproofs_amd.build).  `path`: another build of the
    library (curdleproofs_amd.build.build_variant) beside the default one in the same process — bench.py --ab-lib runs contexts of
    both on the same GPU, pass by pass; every library is loaded once and keeps its own CRS tables."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    if path is not None and os.path.realpath(path) in _libs:
        return _libs[os.path.realpath(path)]
    default = path is None
    path = path or os.environ.get("CPX_LIB", _LIB_PATH)   # CPX_LIB: experimental build variant as THE library of the process
    if not os.path.exists(path):
        raise ImportError("curdleproofs_amd: %s is missing — build it with `python -m curdleproofs_amd.build` "
                          "(hipcc, gfx950). There is no CPU fallback." % _LIB_PATH)
    L = ctypes.CDLL(path)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.cpx_host_alloc.argtypes = [sz]
    L.cpx_host_alloc.restype = vp
    L.cpx_host_free.argtypes = [vp]
    L.cpx_host_free.restype = None
    L.cpx_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.cpx_ctx_destroy.argtypes = [vp]
    L.cpx_ctx_destroy.restype = None
    L.cpx_last_error.argtypes = [vp]
    L.cpx_last_error.restype = ctypes.c_char_p
    L.cpx_device_count.argtypes = []
    L.cpx_ctx_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_longlong]
    L.cpx_ctx_get_option.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)]
    L.cpx_ctx_set_crs.argtypes = [vp, sz, vp, sz]
    L.cpx_crs_sums.argtypes = [vp, vp, vp]
    L.cpx_proof_size.argtypes = [vp]
    L.cpx_proof_size.restype = sz
    L.cpx_batch_size.argtypes = [vp]
    L.cpx_batch_size.restype = sz
    L.cpx_g1_msm.argtypes = [vp, vp, vp, sz, vp]
    L.cpx_g1_msm_jac.argtypes = [vp, vp, vp, sz, vp]
    L.cpx_g1_fold.argtypes = [vp, vp, vp, vp, sz]
    L.cpx_g1_scale.argtypes = [vp, vp, vp, sz, sz, vp]
    L.cpx_g1_normalize.argtypes = [vp, vp, sz, vp, vp]
    L.cpx_g1_decompress.argtypes = [vp, vp, sz, vp, ci]
    L.cpx_g1_decompress_status.argtypes = [vp, vp, sz, vp, ci, vp]
    L.cpx_accum_new.argtypes = [vp, ctypes.POINTER(vp)]
    L.cpx_accum_free.argtypes = [vp]
    L.cpx_accum_free.restype = None
    L.cpx_accum_check.argtypes = [vp, vp, vp, vp, sz, vp]
    L.cpx_accum_verify.argtypes = [vp]
    L.cpx_batch_load.argtypes = [vp, sz, vp, vp, vp, vp, vp]
    L.cpx_batch_load_begin.argtypes = [vp, sz, vp, vp, vp, vp, vp]
    L.cpx_batch_load_end.argtypes = [vp]
    L.cpx_batch_prove.argtypes = [vp, vp, vp, vp, vp, vp]
    L.cpx_batch_verify.argtypes = [vp, vp, vp, vp]
    L.cpx_set_profiling.argtypes = [vp, ci]
    L.cpx_reset_stats.argtypes = [vp]
    L.cpx_get_stat.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double),
                               ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.cpx_batch_verify_fused.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ci)]
    L.cpx_g1_sum_jac.argtypes = [vp, vp, sz, vp, ctypes.POINTER(ci)]
    L.cpx_set_host_threads.argtypes = [vp, ci]
    L.cpx_whisk_generate_shuffle_proof.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.cpx_whisk_is_valid_shuffle_proof.argtypes = [vp, vp, vp, vp, vp, ctypes.POINTER(ci)]
    L.cpx_whisk_generate_tracker_proof.argtypes = [vp, vp, vp, vp, vp]
    L.cpx_whisk_is_valid_tracker_proof.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ci)]
    L.cpx_bench_fpmul.argtypes = [vp, ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
    _libs[os.path.realpath(path)] = L
    if default:
        _lib = L
    return L


def device_count():
    return load_library().cpx_device_count()


def _in(b):
    if isinstance(b, ctypes.Array):   # already marshalled (see Context.marshal): no copy
        return b
    return (ctypes.c_uint8 * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")


def _len(b):
    return ctypes.sizeof(b) if isinstance(b, ctypes.Array) else len(b)


def _out(n):
    return (ctypes.c_uint8 * max(n, 1))()


class _PinnedOwner:
    def __init__(self, L, addr):
        self.L, self.addr = L, addr

    def __del__(self):
        try:
            self.L.cpx_host_free(self.addr)
        except Exception:
            pass


def _pinned_array(ctype, n):
    """ctypes array of n elements in page-locked host memory (falls back to ordinary memory if the allocation fails)."""
    L = load_library()
    nbytes = ctypes.sizeof(ctype) * n
    addr = None if os.environ.get("CPX_PINNED") == "0" else L.cpx_host_alloc(nbytes)   # CPX_PINNED=0: ordinary memory (A/B runs)
    if not addr:
        return (ctype * n)()
    arr = (ctype * n).from_address(addr)
    arr._cpx_owner = _PinnedOwner(L, addr)   # freed when the array object goes away
    return arr


class Context:
    """One HIP device + stream + device-resident CRS (cpx_ctx)."""

    def __init__(self, device=0, options=None, lib=None):
        self._L = load_library(lib)
        h = ctypes.c_void_p()
        rc = self._L.cpx_ctx_create(device, ctypes.byref(h))
        if rc != CPX_OK:
            raise CpxError(rc, "cpx_ctx_create(device=%d): no usable MI355X/HIP device" % device)
        self._h = h
        self.ell = None
        try:
            for key, value in (options or {}).items():
                self.set_option(key, value)
        except Exception:   # an unknown key / a value out of range: do not leak the native context (streams, engine) of the half-built object
            self.close()
            raise

    @property
    def batch(self):
        """instances currently loaded into the engine (cpx_batch_size): the library is the one source of truth — a Whisk shuffle call
        replaces the loaded batch by its single instance, also when it fails half-way"""
        return self._L.cpx_batch_size(self._h) if getattr(self, "_h", None) else 0

    def set_option(self, key, value):
        """Per-context tunable (include/cpx.h cpx_ctx_set_option; keys: curdleproofs_amd/csrc/kernels.h `Options`)."""
        self._check(self._L.cpx_ctx_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = ctypes.c_longlong(0)
        self._check(self._L.cpx_ctx_get_option(self._h, key.encode(), ctypes.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.cpx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != CPX_OK:
            detail = self._L.cpx_last_error(self._h).decode(errors="replace")
            raise (ProofError if rc == CPX_ERR_VERIFY else CpxError)(rc, detail)

    # ---- CRS (crs.rs:37-58) ----
    def set_crs(self, ell, points):
        if len(points) % AFF:
            raise ValueError("CRS points: 96 bytes per affine point")
        # crs.rs:40-42 "not enough points" is reported by the library (CPX_ERR_ARG); surplus points are ignored like the reference's slicing
        self._check(self._L.cpx_ctx_set_crs(self._h, ell, _in(points), len(points) // AFF))
        points = points[:AFF * (ell + 7)]
        self.ell = ell
        self.n = ell + N_BLINDERS
        self.crs_points = bytes(points)

    def crs_sums(self):
        g, h = _out(AFF), _out(AFF)
        self._check(self._L.cpx_crs_sums(self._h, g, h))
        return bytes(g), bytes(h)

    @property
    def proof_size(self):
        return self._L.cpx_proof_size(self._h)

    # ---- tier 0 ----
    def msm(self, bases, scalars):
        """util.rs:19-22; panics in the reference on a length mismatch -> ValueError here."""
        n = len(scalars) // FR
        if len(bases) != AFF * n or len(scalars) != FR * n:
            raise ValueError("number of points != number of scalars")
        o = _out(JAC)
        self._check(self._L.cpx_g1_msm(self._h, _in(bases), _in(scalars), n, o))
        return bytes(o)

    def msm_from_projective(self, bases_jac, scalars):
        n = len(scalars) // FR
        if len(bases_jac) != JAC * n:
            raise ValueError("number of points != number of scalars")
        o = _out(JAC)
        self._check(self._L.cpx_g1_msm_jac(self._h, _in(bases_jac), _in(scalars), n, o))
        return bytes(o)

    def fold(self, PL, PR, gamma):
        half = len(PL) // AFF
        if len(PR) != len(PL) or len(PL) % AFF or len(gamma) != FR:
            raise ValueError("fold: PL / PR must hold the same number of affine points, gamma one scalar")
        b = _in(PL)
        self._check(self._L.cpx_g1_fold(self._h, b, _in(PR), _in(gamma), half))
        return bytes(b)[: AFF * half]

    def scale(self, P, scalars):
        n = len(P) // AFF
        if len(P) % AFF or len(scalars) not in (FR, FR * n):
            raise ValueError("scale: one shared scalar or one scalar per point")
        stride = 0 if len(scalars) == FR else FR
        o = _out(AFF * n)
        self._check(self._L.cpx_g1_scale(self._h, _in(P), _in(scalars), stride, n, o))
        return bytes(o)[: AFF * n]

    def normalize(self, jac, compressed=False):
        n = len(jac) // JAC
        a, c = _out(AFF * n), _out(48 * n)
        self._check(self._L.cpx_g1_normalize(self._h, _in(jac), n, a, c))
        return (bytes(a)[: AFF * n], bytes(c)[: 48 * n]) if compressed else bytes(a)[: AFF * n]

    def decompress(self, comp, check_subgroup=True):
        n = len(comp) // 48
        o = _out(AFF * n)
        self._check(self._L.cpx_g1_decompress(self._h, _in(comp), n, o, 1 if check_subgroup else 0))
        return bytes(o)[: AFF * n]

    def decompress_status(self, comp, check_subgroup=True):
        """per-point form: (affine points, list of status bytes: 0 ok, 1 malformed / not on the curve, 2 not in the subgroup)"""
        n = len(comp) // 48
        o, st = _out(AFF * n), _out(n)
        self._check(self._L.cpx_g1_decompress_status(self._h, _in(comp), n, o, 1 if check_subgroup else 0, st))
        return bytes(o)[: AFF * n], list(bytes(st)[:n])

    # ---- tier 2 ----
    def load_batch(self, vec_R, vec_S, vec_T, vec_U, M):
        """Uploads `batch` instances (concatenated per-proof buffers)."""
        if self.ell is None:
            raise CpxError(CPX_ERR_STATE, "set_crs first")
        batch = len(M) // JAC
        if batch == 0 or len(M) % JAC:
            raise ValueError("M: one Jacobian point (144 bytes) per instance")
        for v in (vec_R, vec_S, vec_T, vec_U):
            if len(v) != batch * self.ell * AFF:
                raise ValueError("every instance vector must hold batch * ell affine points (%d bytes), got %d" % (batch * self.ell * AFF, len(v)))
        self._check(self._L.cpx_batch_load(self._h, batch, _in(vec_R), _in(vec_S), _in(vec_T), _in(vec_U), _in(M)))

    def load_batch_begin(self, vec_R, vec_S, vec_T, vec_U, M):
        """Starts uploading the NEXT batch beside the kernels of the loaded one (cpx_batch_load_begin); the buffers — best page-locked ones from
        `marshal` — must stay untouched until load_batch_end() returns, which makes the staged batch the loaded one."""
        if self.ell is None:
            raise CpxError(CPX_ERR_STATE, "set_crs first")
        batch = len(M) // JAC
        if batch == 0 or len(M) % JAC:
            raise ValueError("M: one Jacobian point (144 bytes) per instance")
        for v in (vec_R, vec_S, vec_T, vec_U):
            if len(v) != batch * self.ell * AFF:
                raise ValueError("every instance vector must hold batch * ell affine points (%d bytes), got %d" % (batch * self.ell * AFF, len(v)))
        self._staged = tuple(_in(v) for v in (vec_R, vec_S, vec_T, vec_U, M))   # kept alive until load_batch_end
        self._check(self._L.cpx_batch_load_begin(self._h, batch, *self._staged))

    def load_batch_end(self):
        try:
            self._check(self._L.cpx_batch_load_end(self._h))
        finally:
            self._staged = None

    @staticmethod
    def marshal(data):
        """Pre-marshal a bytes object (or a list of u32 for permutations) into a page-locked ctypes buffer once (cpx_host_alloc), so
        that repeated batch calls pay neither Python-side copies nor staging copies: the library moves it by asynchronous DMA."""
        if isinstance(data, (bytes, bytearray)):
            buf = _pinned_array(ctypes.c_uint8, max(len(data), 1))
            ctypes.memmove(buf, bytes(data) if len(data) else b"\0", max(len(data), 1))
            return buf
        buf = _pinned_array(ctypes.c_uint32, max(len(data), 1))
        buf[:len(data)] = data
        return buf

    def prove_batch(self, permutations, k, vec_m_blinders, rand, raw=False):
        """CurdleproofsProof::new for every loaded instance; returns a list of serialized proofs (or, with raw=True, a page-locked
        ctypes buffer holding all of them back to back; it is reused by the second-next raw prove of this context)."""
        B, ell, n = self.batch, self.ell, self.n
        if not isinstance(permutations, ctypes.Array):
            if len(permutations) != B * ell:
                raise ValueError("permutations: batch * ell entries")
            permutations = (ctypes.c_uint32 * (B * ell))(*permutations)
        perm = permutations
        if ctypes.sizeof(perm) != 4 * B * ell or _len(k) != B * FR or _len(vec_m_blinders) != B * 4 * FR or _len(rand) != B * (3 * n + 9) * FR:
            raise ValueError("prove_batch: per instance ell u32 permutation entries, k (32 B), 4 blinders, 3n+9 random scalars")
        psz = self.proof_size
        if raw:   # two page-locked output buffers per context, used alternately: a returned buffer stays valid until the prove after the next
            pool = self.__dict__.setdefault("_raw_pool", {})
            slot = pool.get("next", 0)
            if pool.get(("size", slot)) != B * psz:
                pool[("buf", slot)] = _pinned_array(ctypes.c_uint8, B * psz)
                pool[("size", slot)] = B * psz
            out = pool[("buf", slot)]
            pool["next"] = 1 - slot
        else:
            out = _out(B * psz)
        self._check(self._L.cpx_batch_prove(self._h, perm, _in(k), _in(vec_m_blinders), _in(rand), out))
        if raw:
            return out
        blob = bytes(out)
        return [blob[i * psz:(i + 1) * psz] for i in range(B)]

    def verify_batch(self, proofs, rand):
        """CurdleproofsProof::verify for every loaded instance; returns a list of CPX_* verdicts.
        `proofs`: list of serialized proofs, or the raw buffer prove_batch(raw=True) returned."""
        B = self.batch
        psz = self.proof_size
        blob = proofs if isinstance(proofs, ctypes.Array) else b"".join(proofs)
        if _len(blob) != B * psz or _len(rand) != B * 8 * FR:
            raise ValueError("verify_batch: batch * proof_size proof bytes and 8 random factors per proof")
        verdict = (ctypes.c_int * B)(*([CPX_ERR_INTERNAL] * B))   # an entry the library does not write is never read as "accepted"
        self._check(self._L.cpx_batch_verify(self._h, _in(blob), _in(rand), verdict))
        return list(verdict)

    # ---- measurement ----
    def verify_batch_fused_partial(self, proofs, rand):
        """BASELINE config 5: one accumulated MSM over all loaded proofs.  `rand`: 12 Fr per proof.
        Returns (partial sum as a 144-byte Jacobian point, number of structurally invalid proofs)."""
        data = proofs if isinstance(proofs, (bytes, bytearray)) or hasattr(proofs, "_length_") else b"".join(proofs)
        if _len(data) != self.batch * self.proof_size or _len(rand) != self.batch * 12 * FR:
            raise ValueError("verify_batch_fused: batch * proof_size proof bytes and 12 random factors per proof")
        out = _out(JAC)
        bad = ctypes.c_int(0)
        self._check(self._L.cpx_batch_verify_fused(self._h, _in(data), _in(rand), out, ctypes.byref(bad)))
        return bytes(out)[:JAC], bad.value

    def sum_jac(self, points_jac):
        """sum of Jacobian points -> (sum, is_identity)"""
        n = len(points_jac) // JAC
        if len(points_jac) % JAC:
            raise ValueError("sum_jac: 144 bytes per point")
        out = _out(JAC)
        flag = ctypes.c_int(0)
        self._check(self._L.cpx_g1_sum_jac(self._h, _in(points_jac), n, out, ctypes.byref(flag)))
        return bytes(out)[:JAC], bool(flag.value)

    def verify_batch_fused(self, proofs, rand):
        """Single-context form of config 5: True iff every loaded proof is valid (all-or-nothing)."""
        part, bad = self.verify_batch_fused_partial(proofs, rand)
        return bad == 0 and self.sum_jac(part)[1]

    def set_profiling(self, on=True):
        self._check(self._L.cpx_set_profiling(self._h, 1 if on else 0))

    def reset_stats(self):
        self._check(self._L.cpx_reset_stats(self._h))

    def stat(self, name):
        n, ms, by, un = ctypes.c_uint64(0), ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
        self._check(self._L.cpx_get_stat(self._h, name.encode(), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by), ctypes.byref(un)))
        return dict(launches=n.value, ms=ms.value, alg_bytes=by.value, units=un.value)

    KERNELS = ("k_msm_tblw<32, false>", "k_reduce_sets", "k_msm_tblw<16, false>", "k_msm_tblw<8, false>", "k_msm_tblw<4, false>", "k_msm_tblw<2, false>", "k_msm_fix<19, 7>", "k_msm_fix<16, 4>", "k_msm_fix<16, 2>", "k_msm_fix<16, 16>", "k_msm_fix<16, 8>", "k_msm_fix<8, 16>", "k_msm_fix<8, 8>",
               "k_transcript_step1", "k_msm_tblw_pair", "k_late_fix", "k_late_uniform", "k_late_tables", "k_late_msm", "k_finalize_ranges", "k_table_build", "k_msm_accw", "k_msm_tblw<2, true>", "k_msm_tail", "k_smul", "k_finalize", "k_compress", "k_decompress", "host_parallel_for", "host_wait_device", "host_prove_wall", "host_verify_wall")

    def stats(self):
        return {k: self.stat(k) for k in self.KERNELS}

    def set_host_threads(self, t):
        self._check(self._L.cpx_set_host_threads(self._h, t))

    def bench_fpmul(self, blocks=2048, iters=2000, reps=3):
        r = ctypes.c_double(0)
        self._check(self._L.cpx_bench_fpmul(self._h, blocks, iters, reps, ctypes.byref(r)))
        return r.value


class MsmAccumulator:
    """msm_accumulator.rs:22-68.  The random factor is drawn by the caller (the RNG stays outside)."""

    def __init__(self, ctx):
        self._ctx = ctx
        self._L = ctx._L
        h = ctypes.c_void_p()
        ctx._check(self._L.cpx_accum_new(ctx._h, ctypes.byref(h)))
        self._h = h

    def accumulate_check(self, C, vec_x, vec_V, random_factor):
        n = min(len(vec_x) // FR, len(vec_V) // AFF)   # Rust `zip`
        self._ctx._check(self._L.cpx_accum_check(self._h, _in(C), _in(vec_x), _in(vec_V), n, _in(random_factor)))

    def verify(self):
        rc = self._L.cpx_accum_verify(self._h)
        if rc == CPX_ERR_VERIFY:
            raise ProofError(rc)
        self._ctx._check(rc)

    def __del__(self):
        try:
            if self._h:
                self._L.cpx_accum_free(self._h)
                self._h = None
        except Exception:
            pass


class CurdleproofsProof:
    """One serialized proof, with the reference's constructor / verifier names (curdleproofs.rs:59,197)."""

    def __init__(self, data):
        self.data = bytes(data)

    @staticmethod
    def new(ctx, vec_R, vec_S, vec_T, vec_U, M, permutation, k, vec_m_blinders, rand):
        ctx.load_batch(vec_R, vec_S, vec_T, vec_U, M)
        return CurdleproofsProof(ctx.prove_batch(list(permutation), k, vec_m_blinders, rand)[0])

    def verify(self, ctx, vec_R, vec_S, vec_T, vec_U, M, rand):
        ctx.load_batch(vec_R, vec_S, vec_T, vec_U, M)
        v = ctx.verify_batch([self.data], rand)[0]
        if v == CPX_ERR_VERIFY:
            raise ProofError(v)
        if v != CPX_OK:
            raise CpxError(v)

    def serialize(self):
        return self.data

    @staticmethod
    def deserialize(data, log2_n):
        if len(data) != 48 * (18 + 10 * log2_n) + 32 * 7:
            raise CpxError(CPX_ERR_DESERIALIZE, "wrong length")
        return CurdleproofsProof(data)
