#!/usr/bin/env python
"""Summarise rocprofv3 (rocpd SQLite) results as text: per-kernel duration statistics and, when the run
collected PMC counters, per-kernel counter averages.  Usage: tools/rocpd_summary.py results.db [...]

    tools/rocpd_summary.py --bneck-timeline bench_line.json results.db [unprofiled_bench_line.json]

prints instead the TIMELINE of the bottleneck stage from the kernel trace of a bench.py run: for every forward call the
[start, end] interval of each chain's bottleneck stage (first Winograd input transform in front of the chain's first
bottleneck GEMM ... end of its last bottleneck GEMM, per HIP stream / HSA queue), the union of the chains' intervals, and
the average over the calls of bench.py's timed region -- the quantity `roofline.frac` divides by (bench.py measures it with
HIP events on every chain's stream; VERDICT r03 item 1 asks that it reproduce from the trace).  `bench_line.json` is the JSON
line the SAME run printed (steps, warmup, roofline.bneck_union_ms_per_step, roofline.bneck_executed_gflop_per_step)."""
import json
import re
import sqlite3
import sys

FP32_MFMA_PEAK_TFLOPS = 157.3


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    m = re.match(r"(?:void )?(?:eamm::)?(\w+)<(.*)>$", name)
    return name if not m else f"{m.group(1)}<{m.group(2)}>"


def union_ms(intervals):
    """total length of the union of [start, end] intervals (ns) in ms"""
    total, cur0, cur1 = 0, None, None
    for a, b in sorted(intervals):
        if cur1 is None or a > cur1:
            if cur1 is not None:
                total += cur1 - cur0
            cur0, cur1 = a, b
        else:
            cur1 = max(cur1, b)
    if cur1 is not None:
        total += cur1 - cur0
    return total / 1e6


def bottleneck_timeline(con, nres=12, nchains=2):
    """-> list over forward calls of {"chains": [(start, end), ...], "gemm": [(start, end), ...]} (ns), in time order.
    The bottleneck GEMM is the wino4_gemm_kernel instantiation with the most kernel time; a forward call is `nchains` x `nres`
    (= 2 * num_bottleneck_blocks) consecutive dispatches of it in time order (calls never overlap: every call joins its chains
    before it returns); inside a call the chains are told apart by the stream / queue column (eager launches and graph replays
    run on different streams, so the lanes are looked up per call)."""
    cols = [r[1] for r in con.execute("pragma table_info('kernels')")]
    if not cols:   # a view: ask one row
        cols = [d[0] for d in con.execute("select * from kernels limit 1").description]
    t0c, t1c = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    lane_cols = [c for c in ("stream_id", "queue_id", "queue") if c in cols]
    sel = ", ".join(["name", t0c, t1c] + lane_cols)
    rows = con.execute(f"select {sel} from kernels order by {t0c}").fetchall()
    gemm_names = {}   # by total TIME: one-frame launches of the narrow variants (bench.py's latency_b1 leg) are more numerous, far shorter
    for r in rows:
        if "wino4_gemm_kernel" in r[0]:
            gemm_names[r[0]] = gemm_names.get(r[0], 0) + (r[2] - r[1])
    if not gemm_names:
        raise SystemExit("no wino4_gemm_kernel dispatch in the trace (the bottleneck did not run in F(4x4) form)")
    bneck = max(gemm_names, key=gemm_names.get)
    g_all = [k for k, r in enumerate(rows) if r[0] == bneck]
    per_call = nres * nchains
    out, lane_used = [], set()
    prev_end = 0
    for c0 in range(0, len(g_all) - per_call + 1, per_call):
        grp = g_all[c0:c0 + per_call]
        lane_idx = None
        for i, c in enumerate(lane_cols):   # the column that splits this call's dispatches into `nchains` equal groups
            vals = {}
            for k in grp:
                vals[rows[k][3 + i]] = vals.get(rows[k][3 + i], 0) + 1
            if len(vals) == nchains and set(vals.values()) == {nres}:
                lane_idx = 3 + i
                lane_used.add(c)
                break
        chains = {}
        for k in grp:
            chains.setdefault(rows[k][lane_idx] if lane_idx is not None else 0, []).append(k)
        windows = []
        for lane, ks in chains.items():
            start = rows[ks[0]][1]
            # the stage starts with the input transform dispatched on this lane right in front of the chain's first GEMM
            for k in range(ks[0] - 1, -1, -1):
                if rows[k][2] <= prev_end or rows[ks[0]][1] - rows[k][1] > 2_000_000:
                    break
                if (lane_idx is None or rows[k][lane_idx] == lane):
                    if "wino4_input_transform_kernel" in rows[k][0]:
                        start = rows[k][1]
                    break
            windows.append((start, rows[ks[-1]][2]))
        out.append({"chains": sorted(windows), "gemm": [(rows[k][1], rows[k][2]) for k in grp]})
        prev_end = max(rows[k][2] for k in grp)
    return out, short(bneck), ("/".join(sorted(lane_used)) or "one lane")


def timeline_report(bench_json, db):
    line = json.load(open(bench_json))
    roof = line["roofline"]
    con = sqlite3.connect(db)
    calls, kernel, lane = bottleneck_timeline(con, 2 * int(line.get("config", {}).get("num_bottleneck_blocks", 6)), int(roof.get("pass_chains", 2)))
    steps, warm = int(line["steps"]), int(line["warmup"])
    timed = calls[warm:warm + steps] if len(calls) >= warm + steps else calls
    print(f"== bottleneck-stage timeline from {db}")
    print(f"kernel: {kernel}; chains told apart by `{lane}`; {len(calls)} forward calls in the trace, "
          f"calls {warm}..{warm + len(timed) - 1} = bench.py's timed region ({steps} steps after {warm} warm-up)")
    if "graph" in line and len(calls) >= warm + steps + int(line["graph"]["steps"]) + 1:
        # bench.py --graph: the trace ends with [graph warm-up replay, `steps` timed replays, the final check step]
        gs = int(line["graph"]["steps"])
        replays = calls[-(gs + 1):-1]
        print(f"-- graph replays (bench.py --graph: one captured step replayed; no per-launch host cost, so the chains' kernels stay "
              f"together under the profiler as they do in an unprofiled run): calls {len(calls) - gs - 1}..{len(calls) - 2}")
        gu = [union_ms(c["chains"]) for c in replays]
        gsum = [sum((b - a) / 1e6 for a, b in c["chains"]) for c in replays]
        offs = [max(a for a, _ in c["chains"]) - min(a for a, _ in c["chains"]) for c in replays]
        gu_avg = sum(gu) / len(gu)
        gf0 = roof["bneck_executed_gflop_per_step"]
        print(f"   union {gu_avg:.4f} ms (min {min(gu):.4f}, max {max(gu):.4f}), sum of windows {sum(gsum) / len(gsum):.4f} ms, "
              f"offset between the chains' stage starts {sum(offs) / len(offs) / 1e3:.0f} us")
        print(f"   roofline.frac from the replayed trace: {gf0:.2f} GFLOP / {gu_avg:.4f} ms / {FP32_MFMA_PEAK_TFLOPS} = "
              f"{gf0 / gu_avg / FP32_MFMA_PEAK_TFLOPS:.4f}; graph replay ran {line['graph']['value']} frames/s under the profiler")
        if len(sys.argv) > 4:   # the unprofiled run's line: the number the trace has to reproduce
            ref = json.load(open(sys.argv[4]))["roofline"]
            print(f"   unprofiled bench.py (HIP events): bneck_union_ms_per_step {ref['bneck_union_ms_per_step']:.4f}, frac {ref['frac']:.4f}"
                  f" -> replayed trace / unprofiled events = {gu_avg / ref['bneck_union_ms_per_step']:.4f}")
    print("-- eager calls of the timed region (under the profiler the host needs ~1.8 ms to enqueue a chain: the second chain starts late)")
    print(f"{'call':>4s} {'chain windows (ms)':40s} {'union_ms':>9s} {'sum_ms':>9s} {'gemm_union_ms':>13s} {'chain offsets (us)':>20s}")
    un, sm, gu = [], [], []
    for i, c in enumerate(timed):
        w = [(b - a) / 1e6 for a, b in c["chains"]]
        u, g = union_ms(c["chains"]), union_ms(c["gemm"])
        un.append(u); sm.append(sum(w)); gu.append(g)
        offs = " ".join(f"{(a - c['chains'][0][0]) / 1e3:+.0f}" for a, _ in c["chains"])
        print(f"{warm + i:4d} {' '.join(f'{x:.4f}' for x in w):40s} {u:9.4f} {sum(w):9.4f} {g:13.4f} {offs:>20s}")
    n = max(1, len(timed))
    u_avg, s_avg, g_avg = sum(un) / n, sum(sm) / n, sum(gu) / n
    gf = roof["bneck_executed_gflop_per_step"]
    ev = roof["bneck_union_ms_per_step"]
    print(f"average over the timed calls: union {u_avg:.4f} ms, sum of windows {s_avg:.4f} ms, union of the GEMM kernels alone {g_avg:.4f} ms")
    print(f"bench.py (HIP events, same run): bneck_union_ms_per_step {ev:.4f} -> trace / events = {u_avg / ev:.4f}")
    print(f"roofline.frac from the trace: {gf:.2f} GFLOP / {u_avg:.4f} ms / {FP32_MFMA_PEAK_TFLOPS} = {gf / u_avg / FP32_MFMA_PEAK_TFLOPS:.4f}"
          f"   (bench.py line: {roof['frac']:.4f})")
    return u_avg, ev


def per_launch_frac(bench_json, db):
    """`roofline.frac` of the contract line RE-DERIVED from tracked per-kernel averages alone: under rocprofv3 the
    two chains of a 16-frame step drift apart (every dispatch costs the host ~10 us, eager and graph replay alike), so the trace's own
    union of the chains' windows is a different experiment; the per-kernel DURATIONS are not affected by that drift.  Per chain
    (stream) the bottleneck stage is `nres` x (input transform, GEMM) back to back, and the chains run side by side, so

        frac = chains x (executed GFLOP of one GEMM launch) / (avg GEMM duration + avg transform duration) / chip fp32 matrix peak

    with every number taken from the kernel trace (durations) and from the launch geometry the bench line records (FLOP per launch)."""
    line = json.load(open(bench_json))
    roof = line["roofline"]
    chains = int(roof.get("pass_chains", roof.get("chains", 2)))
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('kernels')")] or [d[0] for d in con.execute("select * from kernels limit 1").description]
    # one instantiation may serve several launch sizes (at 512x512 the hourglass encoder's levels run the bottleneck's GEMM instantiation):
    # group by (name, grid) when the trace has the grid, so that the averages are those of ONE launch geometry
    gcols = [c for c in ("grid_size", "grid_size_x", "grid_x", "grid") if c in cols]
    grp = "name" + (", " + gcols[0] if gcols else "")
    rows = con.execute(f"select name, count(*), avg(duration), min(duration), max(duration) from kernels group by {grp}").fetchall()
    gemms = [r for r in rows if "wino4_gemm_kernel" in r[0]]
    trans = [r for r in rows if "wino4_input_transform_kernel" in r[0]]
    if not gemms or not trans:
        raise SystemExit("no wino4 GEMM / input-transform dispatches in the trace")
    # the contract line's kernels = the instantiations with the most TIME (a trace may also hold one-frame launches of the narrow
    # variants -- bench.py's latency_b1 leg -- which are more numerous but short)
    g = max(gemms, key=lambda r: r[1] * r[2])
    t = max(trans, key=lambda r: r[1] * r[2])
    steps_traced = g[1] / (chains * 12.0)
    gflop_launch = roof["per_launch"]["executed_gflop"]
    g_us, t_us = g[2] / 1e3, t[2] / 1e3
    frac_pair = chains * gflop_launch / ((g_us + t_us) * 1e-6) / 1e3 / FP32_MFMA_PEAK_TFLOPS
    frac_gemm = gflop_launch / (g_us * 1e-6) / 1e3 / FP32_MFMA_PEAK_TFLOPS
    print(f"== roofline.frac of the contract line from per-kernel averages of {db}")
    print(f"GEMM       {short(g[0])}: {g[1]} launches (= {steps_traced:.1f} steps x {chains} chains x 12), avg {g_us:.2f} us (min {g[3] / 1e3:.2f}, max {g[4] / 1e3:.2f})")
    print(f"transform  {short(t[0])}: {t[1]} launches, avg {t_us:.2f} us (min {t[3] / 1e3:.2f}, max {t[4] / 1e3:.2f})")
    print(f"executed GFLOP per GEMM launch (bench line, launch geometry): {gflop_launch:.3f}  ({roof['per_launch'].get('frames', '?')} frames per launch)")
    print(f"one launch alone:        {gflop_launch:.3f} GFLOP / {g_us:.2f} us / {FP32_MFMA_PEAK_TFLOPS} TFLOP/s = {frac_gemm:.4f} of the chip "
          f"(a launch occupies 1/{chains} of the CUs: {frac_gemm * chains:.4f} of those)")
    blocks, cus = roof["per_launch"].get("launch_blocks"), roof.get("cus", 256)
    if blocks and blocks * chains > cus:
        # e.g. 512x512 x 8: a 4-frame launch is 256 workgroups -- the chains' launches TIME-share the CUs, so a launch's duration depends on
        # what the other chain runs beside it (min / max above) and "launches side by side on disjoint CUs" does not describe the stage
        print(f"a launch is {blocks} workgroups, the {chains} chains' launches together exceed the {cus} CUs: they time-share the chip and the "
              f"side-by-side derivation below does NOT apply at this size -- the trace's own union of the chains' windows does "
              f"(--bneck-timeline: the profiler's per-dispatch cost is small against this step)")
    print(f"stage, {chains} chains side by side: {chains} x {gflop_launch:.3f} GFLOP / ({g_us:.2f} + {t_us:.2f}) us / {FP32_MFMA_PEAK_TFLOPS} = {frac_pair:.4f}")
    print(f"bench.py line of the same run (HIP-event union of the chains' windows): frac {roof['frac']:.4f}"
          f" -> derived / reported = {frac_pair / roof['frac']:.3f}")
    if len(sys.argv) > 4:
        ref = json.load(open(sys.argv[4]))["roofline"]
        print(f"unprofiled bench.py line: frac {ref['frac']:.4f}, per_launch.avg_launch_ms {ref['per_launch']['avg_launch_ms']:.4f}"
              f" -> derived / unprofiled = {frac_pair / ref['frac']:.3f}; trace GEMM avg / HIP-event GEMM avg = "
              f"{g_us / (ref['per_launch']['avg_launch_ms'] * 1e3):.3f}")


TRAIN_STAGES = (   # (stage, regex on the demangled kernel name) -- first match wins
    ("BatchNorm statistics / apply", r"bn_(nhwc_)?(partial|combine|finalize|apply|local_stats)|bn_nhwc_apply|bn_finalize|bn_combine|bn_nhwc_partial"),
    ("BatchNorm backward", r"bn_(nhwc_)?bwd|bn_bwd"),
    ("weight / bias gradients", r"wgrad|bias_grad|wino4_dy_transform"),
    ("filter packing (weights change every step)", r"pack_dev|conv7_thin_pack|col7_pack"),
    ("convolutions: forward and data gradients", r"wino4_gemm|wino4_input_transform|wino4_output_transform|wino_gemm|wino_input|conv_mfma|conv_col7|"
                                                 r"conv7_thin_in|conv_patch|patch_poly|splitk_reduce|conv_first7|final_shift_sum"),
    ("motion / warp operators (forward + backward)", r"motion_|warp_|kp_prepare|kp_records|antialias|rec_partial|source_to_nhwc|to_u8"),
    ("ATen glue (elementwise / reductions / copies / fills)", r"at::native|rocclr|Cijk_|elementwise|reduce_kernel"),
)


def train_timeline(db, steps_hint=None):
    """A rocprofv3 kernel trace of tools/train_step_bench.py reduced to a per-stage table: kernel time per stage
    in the FORWARD and in the BACKWARD half of a step, launches, and the time the device sits idle between kernels.  Steps are cut at the
    L1 loss: its `abs` kernel ends a forward half, the `sign` kernel of its derivative starts the backward half (one each per step)."""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info('kernels')")] or [d[0] for d in con.execute("select * from kernels limit 1").description]
    t0c, t1c = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = con.execute(f"select name, {t0c}, {t1c} from kernels order by {t0c}").fetchall()
    cuts_f = [i for i, r in enumerate(rows) if "AbsFunctor" in r[0]]
    cuts_b = [i for i, r in enumerate(rows) if "sign_kernel" in r[0]]
    if not cuts_f or len(cuts_f) != len(cuts_b):
        raise SystemExit(f"cannot find the L1 loss's abs / sign kernels in the trace ({len(cuts_f)} / {len(cuts_b)})")
    # step k: forward = (end of step k-1's backward .. abs kernel], backward = [sign kernel .. last kernel before the next forward)
    # the next forward starts at the first kernel after a gap in which the host zeroes the gradients: use the first kernel after the
    # backward's last weight-gradient kernel -- simpler and robust: step k spans rows (bwd_end[k-1], bwd_end[k]] with bwd_end = the last
    # row before the next step's first `kp_prepare` / `motion_front` kernel (the forward's first library kernel)
    firsts = [i for i, r in enumerate(rows) if "kp_prepare" in r[0]]      # the forward's first library kernel
    if len(firsts) < len(cuts_f):
        raise SystemExit("fewer kp_prepare launches than steps in the trace")
    steps = []
    for k, (cf, cb) in enumerate(zip(cuts_f, cuts_b)):
        st = max(i for i in firsts if i < cf)
        nxt = [i for i in firsts if i > cb]
        en = (nxt[0] - 1) if nxt else len(rows) - 1
        steps.append((st, cf, cb, en))
    # the first step pays allocator warm-up (tools/train_step_bench.py drops it too); the last one is followed by the tool's own
    # gradient-norm reductions (ATen), which would count as its backward
    if len(steps) > 2:
        steps = steps[1:-1]
    table = {}
    wall_f = wall_b = idle_f = idle_b = 0.0
    for st, cf, cb, en in steps:
        for lo, hi, half in ((st, cf, "fwd"), (cb, en, "bwd")):
            ivs = [(rows[i][1], rows[i][2]) for i in range(lo, hi + 1)]
            wall = (max(b for _, b in ivs) - min(a for a, _ in ivs)) / 1e6
            busy = union_ms(ivs)
            if half == "fwd":
                wall_f += wall; idle_f += wall - busy
            else:
                wall_b += wall; idle_b += wall - busy
            for i in range(lo, hi + 1):
                name = rows[i][0]
                stage = next((s for s, rx in TRAIN_STAGES if re.search(rx, name)), "other")
                e = table.setdefault(stage, {"fwd": [0.0, 0], "bwd": [0.0, 0]})
                e[half][0] += (rows[i][2] - rows[i][1]) / 1e6
                e[half][1] += 1
    n = len(steps)
    print(f"== training-step timeline from {db}: {n} steps (first and last of the trace dropped), per step")
    print(f"{'stage':58s} {'fwd ms':>8s} {'launches':>9s} {'bwd ms':>8s} {'launches':>9s} {'step ms':>8s} {'share':>6s}")
    tot = sum(v['fwd'][0] + v['bwd'][0] for v in table.values()) + idle_f + idle_b
    for stage, v in sorted(table.items(), key=lambda kv: -(kv[1]['fwd'][0] + kv[1]['bwd'][0])):
        f, b = v["fwd"], v["bwd"]
        print(f"{stage:58s} {f[0] / n:8.3f} {f[1] / n:9.1f} {b[0] / n:8.3f} {b[1] / n:9.1f} {(f[0] + b[0]) / n:8.3f} {100 * (f[0] + b[0]) / tot:5.1f}%")
    print(f"{'idle (no kernel running between first and last kernel)':58s} {idle_f / n:8.3f} {'':9s} {idle_b / n:8.3f} {'':9s} {(idle_f + idle_b) / n:8.3f} {100 * (idle_f + idle_b) / tot:5.1f}%")
    print(f"{'wall clock of the half (first kernel start .. last end)':58s} {wall_f / n:8.3f} {'':9s} {wall_b / n:8.3f} {'':9s} {(wall_f + wall_b) / n:8.3f}")
    return table


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--train-timeline":
        train_timeline(sys.argv[2])
        return
    if len(sys.argv) >= 4 and sys.argv[1] == "--bneck-timeline":
        timeline_report(sys.argv[2], sys.argv[3])
        return
    if len(sys.argv) >= 4 and sys.argv[1] == "--per-launch-frac":
        per_launch_frac(sys.argv[2], sys.argv[3])
        return
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
