#!/usr/bin/env python3
"""Device occupancy from a rocprofv3 kernel trace CSV (--kernel-trace --output-format csv): union of the kernel
intervals vs. the span from first to last kernel, plus the biggest idle gaps and the kernels around them.
Usage: gpu_busy.py <*_kernel_trace.csv> [after_last=<kernel substring>] [from_first=<kernel substring>]
e.g. after_last=k_fix_build from_first=k_compress analyses the bench steps without CRS / batch set-up."""
import csv
import sys


def main(path, after_last=None, from_first=None):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
    rows.sort()
    if after_last:
        idx = [i for i, r in enumerate(rows) if after_last in r[2]]
        if idx:
            rows = rows[idx[-1] + 1:]
    if from_first:
        idx = [i for i, r in enumerate(rows) if from_first in r[2]]
        if idx:
            rows = rows[idx[0]:]
    span = rows[-1][1] - rows[0][0]
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    last_name = rows[0][2]
    for s, e, n in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = n
    busy += cur_e - cur_s
    print("kernels %d  span %.1f ms  busy(union) %.1f ms  idle %.1f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
    tot = {}
    for s_, e_, n_ in rows:
        t = tot.setdefault(n_, [0, 0])
        t[0] += e_ - s_
        t[1] += 1
    print("kernel time (sum of durations; overlapping launches count twice):")
    for n_, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
        print("  %9.2f ms  x%-5d %s" % (t / 1e6, c, n_))
    agg = {}
    for g, a, b in gaps:
        k = (a, b)
        t = agg.setdefault(k, [0, 0])
        t[0] += g
        t[1] += 1
    print("idle time by (kernel before -> kernel after):")
    for (a, b), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print("  %8.2f ms  x%-4d %s -> %s" % (t / 1e6, c, a, b))


if __name__ == "__main__":
    kw = dict(a.split("=", 1) for a in sys.argv[2:])
    main(sys.argv[1], **kw)
