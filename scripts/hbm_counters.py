#!/usr/bin/env python3
"""Merges rocprofv3 --pmc passes (FETCH_SIZE in one run, WRITE_SIZE in another: counter_collection.csv each) into
profiles/<name>.json: per kernel, counter totals (KiB) and dispatch counts, tagged with the bench configuration so
bench.py only uses them for the matching workload.
Usage: hbm_counters.py out.json batch_per_gpu engine_contexts fetch.csv write.csv [commit]"""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")   # keeps the template arguments: k_msm_tblw<64>
            tot[k] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: dict(total=tot[k], dispatches=len(disp[k])) for k in tot}


def main(out, batch, ctxs, fetch_csv, write_csv, commit="unknown"):
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        if k in fe and k in wr:
            res[k] = dict(FETCH_SIZE=fe[k], WRITE_SIZE=wr[k], batch_per_gpu=int(batch), engine_contexts_per_gpu=int(ctxs), commit=commit)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        fd, wd = v["FETCH_SIZE"]["dispatches"], v["WRITE_SIZE"]["dispatches"]
        print("%-28s fetch %10.1f MiB/dispatch (x2 corrected: %10.1f)  write %10.1f MiB/dispatch" % (
            k, v["FETCH_SIZE"]["total"] / fd / 1024, 2 * v["FETCH_SIZE"]["total"] / fd / 1024, v["WRITE_SIZE"]["total"] / wd / 1024))


if __name__ == "__main__":
    main(*sys.argv[1:7])
