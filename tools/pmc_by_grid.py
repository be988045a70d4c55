#!/usr/bin/env python
"""A PMC counter of one kernel family split by LAUNCH SIZE (rocprofv3 rocpd database): per (kernel, workgroups per launch) the number of
launches, the counter's average and the average duration.  tools/pmc_by_grid.py <results.db> <COUNTER> <kernel-name substring>
(the per-kernel tables of tools/rocpd_summary.py average a kernel's launches of every size together)."""
import re
import sqlite3
import sys


def rows(db, counter, needle):
    con = sqlite3.connect(db)
    q = ("select kernel_name, grid_size / workgroup_size, count(*), avg(value), avg(end - start) from counters_collection "
         "where counter_name = ? and kernel_name like ? group by kernel_name, grid_size / workgroup_size order by 1, 2")
    return con.execute(q, (counter, f"%{needle}%")).fetchall()


def main():
    db, counter, needle = sys.argv[1:4]
    print(f"== {counter} by launch size, kernels matching '{needle}' ({db})")
    print(f"{'kernel':58s} {'workgroups':>10s} {'launches':>8s} {'avg ' + counter:>16s} {'avg us':>9s}")
    for name, wgs, n, val, dur in rows(db, counter, needle):
        short = re.sub(r"\(.*\)$", "", name).replace("void ", "").replace("eamm::", "")
        print(f"{short:58s} {wgs:10d} {n:8d} {val:16.1f} {dur / 1e3:9.2f}")


if __name__ == "__main__":
    main()
