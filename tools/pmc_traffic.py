#!/usr/bin/env python
"""Fold two rocprofv3 PMC passes over bench.py (one with FETCH_SIZE, one with WRITE_SIZE -- TCC slots do not fit both,
MI355X_MICROARCH.md "rocprofv3 PMC slots") into profiles/pmc_traffic.json, the table bench.py reads `roofline.traffic`
from.  A record is keyed by (bottleneck form, size, batch) and carries the digest of the dominant kernel's source file,
so bench.py can tell whether the kernel it is running is the one that was profiled.

    python tools/pmc_traffic.py <fetch_results.db> <write_results.db> <size> <batch> [label] [out.json] [chains]

Run on the GPU box after tools/gpu_profile.sh (which makes the two passes); commit the json with the summaries.
"""
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles", "pmc_traffic.json")
DOMINANT = {4: ("wino4_gemm_kernel", "conv_winograd4.hip"), 2: ("wino_gemm_kernel", "conv_winograd.hip"),
            0: ("conv_mfma_dma_kernel<3, 3", "conv_mfma_dma.hip")}


def counters(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
                       "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {re.sub(r"\(.*\)$", "", k).replace("void ", "").replace("eamm::", ""): (n, v, d / 1e3) for k, n, v, d in rows}


def main():
    fetch_db, write_db, size, batch = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    label = sys.argv[5] if len(sys.argv) > 5 else ""
    out = sys.argv[6] if len(sys.argv) > 6 else OUT   # on the GPU box only gpurun_out/ travels back: write there, copy later
    chains = int(sys.argv[7]) if len(sys.argv) > 7 else 1   # bottleneck chains of the profiled run (frames per launch = batch / chains)
    fetch, write = counters(fetch_db, "FETCH_SIZE"), counters(write_db, "WRITE_SIZE")
    table = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            table = json.load(f)
    for form, (prefix, src) in DOMINANT.items():
        names = [k for k in fetch if k.startswith(prefix) and k in write]
        if not names:
            continue
        name = max(names, key=lambda k: fetch[k][0] * fetch[k][2])   # the variant that took the most time
        with open(os.path.join(ROOT, "eamm_amd", "csrc", src), "rb") as f:
            digest = hashlib.sha256(f.read()).hexdigest()[:16]
        table[f"form{form}_{size}x{size}_b{batch}_c{chains}"] = {
            "frames_per_launch": batch // chains if form != 0 else None,
            "kernel": name, "launches_profiled": fetch[name][0],
            "fetch_size_kb": round(fetch[name][1], 1), "write_size_kb": round(write[name][1], 1),
            "hbm_bytes_per_launch": int((fetch[name][1] * 2 + write[name][1]) * 1024),
            "avg_launch_us_under_pmc": round(fetch[name][2], 2),
            "note": "FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section); "
                    "WRITE_SIZE as reported; both averaged over the launches of `bench.py` itself under rocprofv3 --pmc",
            "source_sha256_16": digest, "source_file": "eamm_amd/csrc/" + src, "files": label,
        }
        print(f"form {form}: {name}  FETCH {fetch[name][1]:.0f} KB x2  WRITE {write[name][1]:.0f} KB  ({fetch[name][0]} launches)")
    # every other kernel of the step, for DESIGN.md's traffic table
    table[f"all_kernels_{size}x{size}_b{batch}"] = {
        k: {"launches": fetch[k][0], "fetch_size_kb": round(fetch[k][1], 1), "write_size_kb": round(write.get(k, (0, 0, 0))[1], 1),
            "avg_us_under_pmc": round(fetch[k][2], 2)} for k in sorted(fetch) if "rocclr" not in k}
    with open(out, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    main()
