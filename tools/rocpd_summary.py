#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/profile_bench.sh into one text file fit
for profiles/: per-kernel duration stats and per-kernel PMC counter averages.
Usage: python tools/rocpd_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def main(root):
    for sub in sorted(os.listdir(root)):
        dbs = glob.glob(os.path.join(root, sub, "*.db"))
        if not dbs:
            continue
        db = dbs[0]
        print("== %s (%s)" % (sub, os.path.basename(db)))
        if sub == "stats":
            print("kernel, calls, total_us, avg_us, min_us, max_us")
            rows = q(db, "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                         "max(end-start)/1e3 from kernels group by name order by 3 desc")
            for r in rows:
                print("%s, %d, %.1f, %.1f, %.1f, %.1f" % r)
        else:
            print("kernel, counter, dispatches, avg_value")
            rows = q(db, "select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                         "group by kernel_name, counter_name order by 1, 2")
            for r in rows:
                print("%s, %s, %d, %.6g" % r)
        print()
    for name in ("stats.json", "pmc_write.json", "pmc_fetch.json", "pmc_sq.json"):
        path = os.path.join(root, name)
        if os.path.exists(path):
            print("== bench.py line of the %s run" % name.split(".")[0])
            print(open(path).read().strip())
            print()


if __name__ == "__main__":
    main(sys.argv[1])
