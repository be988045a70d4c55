"""tools/rocpd_summary.py --bneck-timeline: the union of the chains' bottleneck windows re-derived from a rocprofv3 kernel
trace (`roofline.frac` must reproduce from profiles/).  CPU test on a synthetic rocpd-shaped database:
two chains on two streams, known offsets."""
import json
import os
import sqlite3
import subprocess
import sys

from conftest import ROOT

GEMM = "void eamm::wino4_gemm_kernel<2, 4, 2, 4, 2, 8, 0>(eamm::Wino4Args)"
GEMM_HG = "void eamm::wino4_gemm_kernel<2, 4, 2, 4, 2, 4, 0>(eamm::Wino4Args)"
TR = "void eamm::wino4_input_transform_kernel<4>(float const*, float const*, float const*, int, int, int, int, float*)"


def test_bottleneck_timeline_union(tmp_path):
    db = tmp_path / "kt_results.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer, duration integer, stream_id integer, queue_id integer, "
                "vgpr_count integer, accum_vgpr_count integer, lds_size integer)")
    rows = []
    calls, nres, T, G = 5, 12, 30_000, 170_000          # ns
    for call in range(calls):
        base = call * 5_000_000
        for lane, off in ((1, 0), (2, 40_000)):          # the second chain starts 40 us later
            t = base + off
            rows.append(("eamm::warp_features_kernel", t - 25_000, t - 5_000, 20_000, lane, lane))
            rows.append((GEMM_HG, t - 600_000, t - 560_000, 40_000, lane, lane))   # an hourglass level: not the bottleneck
            for _ in range(nres):
                rows.append((TR, t, t + T, T, lane, lane))
                rows.append((GEMM, t + T, t + T + G, G, lane, lane))
                t += T + G
    # bench.py --graph: [graph warm-up replay, 2 timed replays, final eager check step]; replays run on the graph's own streams
    for call, lanes in ((calls, (7, 8)), (calls + 1, (7, 8)), (calls + 2, (7, 8)), (calls + 3, (1, 2))):
        base = call * 5_000_000
        for lane, off in zip(lanes, (0, 5_000)):
            t = base + off
            for _ in range(nres):
                rows.append((TR, t, t + T, T, lane, lane))
                rows.append((GEMM, t + T, t + T + G, G, lane, lane))
                t += T + G
    # one-frame launches of the narrow variants (bench.py's latency_b1 leg): more numerous than the contract line's, far less time
    NARROW = "void eamm::wino4_gemm_kernel<2, 2, 2, 4, 2, 4, 0>(eamm::Wino4Args)"
    TR1 = TR.replace("<4>", "<1>")
    t = (calls + 10) * 5_000_000
    for _ in range(400):
        rows.append((TR1, t, t + 8_000, 8_000, 1, 1))
        rows.append((NARROW, t + 8_000, t + 29_000, 21_000, 1, 1))
        t += 40_000
    con.executemany("insert into kernels values (?,?,?,?,?,?,124,0,131072)", rows)
    con.commit()
    con.close()
    window = nres * (T + G) / 1e6
    line = {"steps": 3, "warmup": 1, "roofline": {"bneck_union_ms_per_step": window + 0.040, "frac": 0.59, "pass_chains": 2,
                                                  "bneck_executed_gflop_per_step": 231.93},
            "graph": {"steps": 2, "value": 3800.0}}
    (tmp_path / "line.json").write_text(json.dumps(line))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), "--bneck-timeline",
                          str(tmp_path / "line.json"), str(db)], capture_output=True, text=True, check=True).stdout
    assert "wino4_gemm_kernel<2, 4, 2, 4, 2, 8, 0>" in out and "stream_id" in out and "9 forward calls" in out
    replay = [l for l in out.splitlines() if l.strip().startswith("union ")][0]
    assert f"union {window + 0.005:.4f} ms" in replay and "offset between the chains' stage starts 5 us" in replay
    avg = [l for l in out.splitlines() if l.startswith("average over the timed calls")][0]
    assert f"union {window + 0.040:.4f} ms" in avg and f"sum of windows {2 * window:.4f} ms" in avg
    assert "trace / events = 1.0000" in out
    # --per-launch-frac (round 6): roofline.frac from the trace's per-kernel AVERAGES alone -- 2 chains x 9.664 GFLOP
    # per launch / (170 us GEMM + 30 us transform) / 157.3 TFLOP/s; the hourglass-level GEMM (fewer dispatches) must not be picked
    line["roofline"]["per_launch"] = {"executed_gflop": 9.664, "avg_launch_ms": 0.170, "frames": 8}
    (tmp_path / "line.json").write_text(json.dumps(line))
    out3 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), "--per-launch-frac",
                           str(tmp_path / "line.json"), str(db), str(tmp_path / "line.json")], capture_output=True, text=True, check=True).stdout
    want = 2 * 9.664 / ((G + T) * 1e-9) / 1e3 / 157.3     # GFLOP / s -> TFLOP/s -> fraction of the peak
    assert f"= {want:.4f}" in out3 and "wino4_gemm_kernel<2, 4, 2, 4, 2, 8, 0>" in out3 and "avg 170.00 us" in out3 and "avg 30.00 us" in out3
    assert f"derived / reported = {want / 0.59:.3f}" in out3 and "trace GEMM avg / HIP-event GEMM avg = 1.000" in out3
    # the plain summary mode still works on the same database
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), str(db)], capture_output=True, text=True,
                          check=True).stdout
    assert "wino4_gemm_kernel<2, 4, 2, 4, 2, 8, 0>" in out2 and "total_ms" in out2


def test_pmc_counter_split_by_launch_size(tmp_path):
    """tools/pmc_by_grid.py: one kernel launched at two sizes (the contract line's 8-frame launches, the clip leg's 32-frame launches)
    keeps two rows -- rocpd_summary's per-kernel table would average them (profiles/r05_experiments.txt 14)."""
    db = tmp_path / "pmc_results.db"
    con = sqlite3.connect(db)
    con.execute("create table counters_collection (kernel_name text, grid_size integer, workgroup_size integer, counter_name text, "
                "value real, start integer, end integer)")
    rows = [(GEMM, 128 * 512, 512, "FETCH_SIZE", 82000.0 + i, 0, 164_000) for i in range(3)]
    rows += [(GEMM, 512 * 512, 512, "FETCH_SIZE", 254000.0, 0, 336_000) for _ in range(5)]
    rows += [(GEMM, 512 * 512, 512, "WRITE_SIZE", 1.0, 0, 1), (TR, 1024 * 256, 256, "FETCH_SIZE", 15000.0, 0, 20_000)]
    con.executemany("insert into counters_collection values (?,?,?,?,?,?,?)", rows)
    con.commit()
    con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_by_grid.py"), str(db), "FETCH_SIZE", "wino4_gemm_kernel"],
                         capture_output=True, text=True, check=True).stdout.splitlines()
    body = [l.split() for l in out[2:]]
    assert len(body) == 2 and "input_transform" not in "".join(out)
    assert body[0][-4:] == ["128", "3", "82001.0", "164.00"] and body[1][-4:] == ["512", "5", "254000.0", "336.00"]


def test_train_step_timeline_by_stage(tmp_path):
    """tools/rocpd_summary.py --train-timeline (round 6): a kernel trace of tools/train_step_bench.py reduced to kernel
    time per stage in the forward and backward halves of a step + idle time; steps cut at the L1 loss's abs / sign kernels."""
    db = tmp_path / "train.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer, duration integer)")
    rows, t = [], [0]

    def add(name, dur, gap=0):
        t[0] += gap
        rows.append((name, t[0], t[0] + dur, dur))
        t[0] += dur
    for step in range(4):
        add("void eamm::kp_prepare_kernel(float const*)", 5_000, 100_000)
        add(GEMM, 100_000)
        add("void eamm::bn_nhwc_partial_kernel(float const*)", 20_000, 1_000)
        add("void at::native::vectorized_elementwise_kernel<4, at::native::AbsFunctor<float>, std::array<char*, 2ul> >(int)", 5_000)
        add("void at::native::reduce_kernel<512, 1, at::native::ReduceOp<float, at::native::MeanOps<float> > >(int)", 5_000)
        add("void at::native::vectorized_elementwise_kernel<4, at::native::sign_kernel_cuda(at::TensorIteratorBase&)>(int)", 5_000)
        add("void eamm::conv_wgrad_kernel<32>(eamm::WgradArgs)", 200_000)
        add("void eamm::bn_nhwc_bwd_apply_kernel<false>(float const*)", 30_000, 2_000)
    con.executemany("insert into kernels values (?,?,?,?)", rows)
    con.commit()
    con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), "--train-timeline", str(db)], capture_output=True,
                         text=True, check=True).stdout
    assert "2 steps" in out
    line = {l.split("  ")[0]: l.split() for l in out.splitlines()[2:]}
    assert line["weight / bias gradients"][-6:] == ["0.000", "0.0", "0.200", "1.0", "0.200", "54.3%"]
    assert line["convolutions: forward and data gradients"][-6:-1] == ["0.100", "1.0", "0.000", "0.0", "0.100"]
    assert line["BatchNorm backward"][-4:-1] == ["0.030", "1.0", "0.030"] and line["BatchNorm statistics / apply"][-6:-4] == ["0.020", "1.0"]
    idle = [l for l in out.splitlines() if l.startswith("idle")][0].split()
    assert idle[-4:-1] == ["0.001", "0.002", "0.003"]
