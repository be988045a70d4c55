#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd SQLite) results as text: per-kernel duration statistics and, when the run
collected PMC counters, per-kernel counter averages.  Usage: tools/rocpd_summary.py results.db [...]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    m = re.match(r"(?:void )?(?:eamm::)?(\w+)<(.*)>$", name)
    return name if not m else f"{m.group(1)}<{m.group(2)}>"


def main():
    for path in sys.argv[1:]:
        con = sqlite3.connect(path)
        print(f"== {path}")
        rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                           "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name "
                           "order by sum(duration) desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        print(f"{'kernel':78s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} vgpr agpr lds")
        for n, c, tot, avg, mn, mx, vg, ag, lds in rows:
            print(f"{short(n)[:78]:78s} {c:6d} {tot/1e6:9.3f} {avg/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f} {vg} {ag} {lds}")
        try:
            pmc = con.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from "
                              "counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
        except sqlite3.Error:
            pmc = []
        if pmc:
            print(f"-- PMC (average per dispatch)")
            for kn, cn, c, v, d in pmc:
                if "rocclr" in kn:
                    continue
                print(f"{short(kn)[:60]:60s} {cn:28s} n={c:4d} avg={v:16.1f} avg_dur_us={d/1e3:9.2f}")


if __name__ == "__main__":
    main()
