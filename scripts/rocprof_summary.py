#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table:
calls, total ms, avg us, min/max us, share.  Usage: rocprof_summary.py results.db [out.md]"""
import sqlite3
import sys


def summarise(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
    scol = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    q = ("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) from %s d join %s s "
         "on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, disp, sym, name_col))
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for name, calls, tot, mn, mx in rows:
        short = name.split("(")[0]
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (short, calls, tot / 1e6, tot / calls / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    return "\n".join(lines)


if __name__ == "__main__":
    out = summarise(sys.argv[1])
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(out + "\n")
    print(out)
