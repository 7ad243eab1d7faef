"""Kernel-time breakdown of ONE ell = 252 verification alone on the GPU (host-driven path), like single_proof_breakdown.py for the prover."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curdleproofs_amd as cpx
from tests.oracle_lib import Oracle
orc = Oracle()
ell = 252
crs = orc.generate_crs_points(ell); inst = orc.make_instance(ell, 0, crs)
ctx = cpx.Context(0); ctx.set_crs(ell, crs)
ctx.load_batch(inst["vec_R"], inst["vec_S"], inst["vec_T"], inst["vec_U"], inst["M"])
for rep in range(3):
    v = ctx.verify_batch([inst["proof"]], inst["verifier_rand"])
ctx.set_profiling(True); ctx.reset_stats()
t0 = time.perf_counter()
v = ctx.verify_batch([inst["proof"]], inst["verifier_rand"])
t1 = time.perf_counter()
print("verify %.1f ms" % ((t1 - t0) * 1e3), v)
for k, st in sorted(ctx.stats().items(), key=lambda kv: -kv[1]["ms"]):
    if st["launches"]: print("%-22s %7.2f ms x%d" % (k, st["ms"], st["launches"]))
