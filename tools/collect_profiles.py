#!/usr/bin/env python
"""Copy the text/json summaries of one gpurun round (tools/gpu_round.sh <tag>) from gpurun_out/ (scratch) into profiles/
(tracked), and merge the per-size pmc_traffic.json tables into profiles/pmc_traffic.json.   tools/collect_profiles.py <tag> [round]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r06"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
table = {}
if os.path.exists(os.path.join(P, "pmc_traffic.json")):
    table = json.load(open(os.path.join(P, "pmc_traffic.json")))
for size_tag in ("256_b16", "512_b8"):
    d = os.path.join(G, f"prof_{rnd}{tag}_{size_tag}")
    if not os.path.isdir(d):
        continue
    for src, dst in (("kernel_trace_stats.txt", f"{rnd}_{size_tag}_rocprofv3_kernel_trace_stats.txt"),
                     ("bench_under_kernel_trace.json", f"{rnd}_{size_tag}_bench_under_kernel_trace.json"),
                     ("bneck_timeline.txt", f"{rnd}_{size_tag}_bneck_timeline.txt"),
                     ("per_launch_frac.txt", f"{rnd}_{size_tag}_per_launch_frac.txt"),
                     ("pmc_fetch.txt", f"{rnd}_{size_tag}_pmc_fetch.txt"), ("pmc_write.txt", f"{rnd}_{size_tag}_pmc_write.txt"),
                     ("pmc_sq.txt", f"{rnd}_{size_tag}_pmc_sq.txt"), ("pmc_lds.txt", f"{rnd}_{size_tag}_pmc_lds.txt")):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(P, dst))
    t = os.path.join(d, "pmc_traffic.json")
    if os.path.exists(t):   # each run's table starts from the committed one: take only the records of the run's own size
        size = size_tag.split("_")[0]
        table.update({k: v for k, v in json.load(open(t)).items() if f"_{size}x{size}_" in k})
r = os.path.join(G, f"{rnd}_{tag}")
for src, dst in (("bench_256_b16.json", f"{rnd}_bench_256_b16.json"), ("bench_512_b8.json", f"{rnd}_bench_512_b8.json"),
                 ("module_latency.txt", f"{rnd}_module_latency.txt"), ("bn_bench.txt", f"{rnd}_bn_bench.txt"),
                 ("backward_bench.txt", f"{rnd}_backward_bench.txt"), ("train_step.txt", f"{rnd}_train_step.txt")):
    if os.path.exists(os.path.join(r, src)):
        shutil.copy(os.path.join(r, src), os.path.join(P, dst))
json.dump(table, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print("profiles/ updated from round", tag, "-- traffic records:", sorted(k for k in table if k.startswith("form")))
