"""Kernel timeline of ONE lone-proof prove (and verify) from a rocprofv3 kernel trace.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python <repo>/scripts/single_proof_timeline.py run
  python <repo>/scripts/single_proof_timeline.py report /tmp/tl
The run leg proves 6 times and verifies 6 times with pauses between the calls; the report leg prints the last call of each kind:
start offset, duration and the idle gap before every kernel."""
import sys, os, time, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run():
    import curdleproofs_amd as cpx
    from tests.oracle_lib import Oracle
    orc = Oracle()
    ell = 252
    crs = orc.generate_crs_points(ell); inst = orc.make_instance(ell, 0, crs)
    ctx = cpx.Context(0); ctx.set_crs(ell, crs)
    ctx.load_batch(inst["vec_R"], inst["vec_S"], inst["vec_T"], inst["vec_U"], inst["M"])
    for rep in range(6):
        time.sleep(0.05)
        t0 = time.perf_counter()
        proofs = ctx.prove_batch(inst["permutation"], inst["k"], inst["vec_m_blinders"], inst["prover_rand"])
        t1 = time.perf_counter()
        time.sleep(0.05)
        t2 = time.perf_counter()
        v = ctx.verify_batch(proofs, inst["verifier_rand"])
        t3 = time.perf_counter()
        print("prove %.2f ms verify %.2f ms" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3), proofs[0] == inst["proof"], v)

def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("cpx::", "").replace("void ", "")) for r in csv.DictReader(open(f))]
    rows.sort()
    # calls are separated by >= 30 ms of idle time
    groups, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - cur[-1][1] > 30e6:
            groups.append(cur); cur = [r]
        else:
            cur.append(r)
    groups.append(cur)
    for name, grp in (("prove", groups[-2]), ("verify", groups[-1])):
        t0 = grp[0][0]
        busy = sum(e - s for s, e, _ in grp)
        print("== last %s: %d kernels, span %.2f ms, kernel time %.2f ms" % (name, len(grp), (grp[-1][1] - t0) / 1e6, busy / 1e6))
        prev = t0
        for s, e, k in grp:
            print("  +%8.1f us  %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, k[:70]))
            prev = max(prev, e)

if __name__ == "__main__":
    if sys.argv[1] == "run": run()
    else: report(sys.argv[2])
