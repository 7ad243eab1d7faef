"""Prove + verify passes per second at a given batch size (one engine context), synthetic ell = 252 instances: the loop behind
bench.py's `baseline_shapes`, as a stand-alone script for tuning the launch heuristics (CPX_* knobs are read from the environment).
Usage: python scripts/shape_sweep.py <batch> [ell]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import curdleproofs_amd as cpx
import bench

B = int(sys.argv[1])
ell = int(sys.argv[2]) if len(sys.argv) > 2 else 252
c = cpx.Context(0)
cache = "/tmp/shape_%d_%d.pkl" % (ell, B)
import pickle
if os.path.exists(cache):
    data = pickle.load(open(cache, "rb"))
    c.set_crs(ell, data["crs"])
else:
    data = bench.synth_batch(c, ell, B, seed=77)
    pickle.dump(data, open(cache, "wb"))
c.load_batch(data["vec_R"], data["vec_S"], data["vec_T"], data["vec_U"], data["M"])
a = [c.marshal(data[k]) for k in ("perms", "k", "mbl", "prand", "vrand")]
pr = c.prove_batch(a[0], a[1], a[2], a[3], raw=True)
ok = all(v == 0 for v in c.verify_batch(pr, a[4]))
reps, t0 = 0, time.perf_counter()
tp = tv = 0.0
while reps < 3 or time.perf_counter() - t0 < 1.5:
    t1 = time.perf_counter()
    pr = c.prove_batch(a[0], a[1], a[2], a[3], raw=True)
    t2 = time.perf_counter()
    ok &= all(v == 0 for v in c.verify_batch(pr, a[4]))
    t3 = time.perf_counter()
    tp += t2 - t1
    tv += t3 - t2
    reps += 1
dt = time.perf_counter() - t0
print("B=%d ell=%d: %.0f proofs/s, %.1f ms per pass (prove %.1f + verify %.1f), ok=%s  [%s]" % (
    B, ell, B * reps / dt, 1e3 * dt / reps, 1e3 * tp / reps, 1e3 * tv / reps, ok,
    " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("CPX_"))))
if os.environ.get("SWEEP_STATS"):   # per-kernel times of one more pass (event timing serialises the kernels: shares, not wall time)
    c.set_profiling(True); c.reset_stats()
    pr = c.prove_batch(a[0], a[1], a[2], a[3], raw=True)
    c.verify_batch(pr, a[4])
    for k, v in sorted(c.stats().items(), key=lambda kv: -kv[1]["ms"]):
        if v["launches"]: print("  %-24s %8.2f ms x%d" % (k, v["ms"], v["launches"]))
