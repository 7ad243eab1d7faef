"""Kernel timeline of one prove + verify pass at a given batch size (one context) from a rocprofv3 kernel trace.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python <repo>/scripts/batch_timeline.py run <B>
  python <repo>/scripts/batch_timeline.py report /tmp/tl"""
import sys, os, time, glob, csv, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run(B, ell=252):
    import curdleproofs_amd as cpx
    import bench
    c = cpx.Context(0)
    data = bench.synth_batch(c, ell, B, seed=77)
    c.load_batch(data["vec_R"], data["vec_S"], data["vec_T"], data["vec_U"], data["M"])
    a = [c.marshal(data[k]) for k in ("perms", "k", "mbl", "prand", "vrand")]
    for rep in range(4):
        time.sleep(0.1)
        t0 = time.perf_counter()
        pr = c.prove_batch(a[0], a[1], a[2], a[3], raw=True)
        t1 = time.perf_counter()
        time.sleep(0.1)
        t2 = time.perf_counter()
        v = c.verify_batch(pr, a[4])
        t3 = time.perf_counter()
        print("prove %.2f ms verify %.2f ms" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3), all(x == 0 for x in v))

def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("cpx::", "").replace("void ", "")) for r in csv.DictReader(open(f))]
    rows.sort()
    groups, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[0] - cur[-1][1] > 60e6:
            groups.append(cur); cur = [r]
        else:
            cur.append(r)
    groups.append(cur)
    for name, grp in (("prove", groups[-2]), ("verify", groups[-1])):
        t0 = grp[0][0]
        print("== last %s: %d kernels, span %.2f ms, kernel time %.2f ms" % (name, len(grp), (grp[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in grp) / 1e6))
        prev = t0
        for s, e, k in grp:
            print("  +%8.1f us  %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, k[:70]))
            prev = max(prev, e)

if __name__ == "__main__":
    if sys.argv[1] == "run": run(int(sys.argv[2]))
    else: report(sys.argv[2])
