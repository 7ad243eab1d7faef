"""Prove + verify passes of B proofs split over K engine contexts (one host thread each) on one GPU: does a mid-size batch finish sooner as
K interleaved sub-batches (one context's latency-bound step kernels beside another's MSM waves)?
Usage: python scripts/shape_ctxs.py <batch> <contexts> [ell]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curdleproofs_amd as cpx
import bench

B, K = int(sys.argv[1]), int(sys.argv[2])
ell = int(sys.argv[3]) if len(sys.argv) > 3 else 252
ctxs = [cpx.Context(0) for _ in range(K)]
cache = "/tmp/shape_%d_%d.pkl" % (ell, B)
import pickle
if os.path.exists(cache):
    data = pickle.load(open(cache, "rb"))
    ctxs[0].set_crs(ell, data["crs"])
else:
    data = bench.synth_batch(ctxs[0], ell, B, seed=77)
    pickle.dump(data, open(cache, "wb"))
Bk, n_ = B // K, ell + 4
AFF, FR, JAC = cpx.AFF, cpx.FR, cpx.JAC
args = []
for k, c in enumerate(ctxs):
    if k:
        c.set_crs(ell, data["crs"])
    cut = lambda buf, unit: buf[k * Bk * unit:(k + 1) * Bk * unit]
    c.load_batch(cut(data["vec_R"], ell * AFF), cut(data["vec_S"], ell * AFF), cut(data["vec_T"], ell * AFF), cut(data["vec_U"], ell * AFF), cut(data["M"], JAC))
    args.append([c.marshal(x) for x in (data["perms"][k * Bk * ell:(k + 1) * Bk * ell], cut(data["k"], FR), cut(data["mbl"], 4 * FR), cut(data["prand"], (3 * n_ + 9) * FR),
                                        cut(data["vrand"], 8 * FR))])
ok = [True] * K


def one_pass(k):
    c, a = ctxs[k], args[k]
    pr = c.prove_batch(a[0], a[1], a[2], a[3], raw=True)
    ok[k] &= all(v == 0 for v in c.verify_batch(pr, a[4]))


def pass_all():
    th = [threading.Thread(target=one_pass, args=(k,)) for k in range(1, K)]
    for t in th:
        t.start()
    one_pass(0)
    for t in th:
        t.join()


pass_all()
reps, t0 = 0, time.perf_counter()
while reps < 3 or time.perf_counter() - t0 < 1.5:
    pass_all()
    reps += 1
dt = time.perf_counter() - t0
print("B=%d over %d contexts, ell=%d: %.0f proofs/s, %.2f ms per pass, ok=%s" % (B, K, ell, B * reps / dt, 1e3 * dt / reps, all(ok)))
