import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curdleproofs_amd as cpx
from tests.oracle_lib import Oracle
import bench
orc = Oracle()
ell, B = int(sys.argv[1]), int(sys.argv[2])
n = ell + 4
c = cpx.Context(0)
data = bench.synth_batch(c, ell, B, seed=1234)
AFF, FR, JAC = cpx.AFF, cpx.FR, cpx.JAC
c.load_batch(data["vec_R"], data["vec_S"], data["vec_T"], data["vec_U"], data["M"])
proofs = c.prove_batch(data["perms"], data["k"], data["mbl"], data["prand"])
def oracle(i):
    return orc.prove(ell, data["crs"], data["vec_R"][i*ell*AFF:(i+1)*ell*AFF], data["vec_S"][i*ell*AFF:(i+1)*ell*AFF], data["vec_T"][i*ell*AFF:(i+1)*ell*AFF],
                     data["vec_U"][i*ell*AFF:(i+1)*ell*AFF], data["M"][i*JAC:(i+1)*JAC], data["perms"][i*ell:(i+1)*ell], data["k"][i*FR:(i+1)*FR],
                     data["mbl"][i*4*FR:(i+1)*4*FR], data["prand"][i*(3*n+9)*FR:(i+1)*(3*n+9)*FR])
for i in (0, 1, B - 1):
    o = oracle(i)
    g = proofs[i]
    print("proof", i, "equal" if o == g else "DIFF at byte %d" % next(k for k in range(len(o)) if o[k] != g[k]))
print("verify", all(v == 0 for v in c.verify_batch(proofs, data["vrand"])))
