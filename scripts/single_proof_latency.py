import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curdleproofs_amd as cpx
from tests.oracle_lib import Oracle
orc = Oracle()
for ell in (252, 1020):
    t0 = time.time(); crs = orc.generate_crs_points(ell); inst = orc.make_instance(ell, 0, crs); print("oracle instance", ell, round(time.time() - t0, 2), "s")
    ctx = cpx.Context(0)
    t0 = time.time(); ctx.set_crs(ell, crs); print("set_crs", round(time.time() - t0, 2), "s")
    ctx.load_batch(inst["vec_R"], inst["vec_S"], inst["vec_T"], inst["vec_U"], inst["M"])
    for rep in range(4):
        t0 = time.perf_counter()
        proofs = ctx.prove_batch(inst["permutation"], inst["k"], inst["vec_m_blinders"], inst["prover_rand"])
        t1 = time.perf_counter()
        v = ctx.verify_batch(proofs, inst["verifier_rand"])
        t2 = time.perf_counter()
        print(ell, "prove %.1f ms  verify %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), proofs[0] == inst["proof"], v)
    del ctx
