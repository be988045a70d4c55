"""The launch plan BASELINE.json configs[3] is quoted on (VERDICT r04 "What's weak" 1): calls of 64 frames at 256x256 run as
FOUR whole-pass chains on the caller's stream + three side streams taken from a per-device pool shared by every handle
(eamm_api.hip: pass_chains, StreamLease).  An 8-GPU job over a 2048-frame clip gives each rank 256 frames = four such calls.

Checked here, through the C ABI: (a) a 64-frame call against the same frames in four 16-frame calls (two chains of 8: the
plan `value` is measured on) and against the CPU oracle on frames {0, 17, 40, 63}, every output key; (b) ragged calls of 50
and 33 frames (two chains of unequal length on the four-chain handle); (c) two handles on two caller streams driven
concurrently -- from one host thread and from two -- equal to their serial results; (d) HIP-graph capture of the 64-frame call
replays bit-exactly, runs on the handle's private streams and never touches the pool while another handle uses it;
(e) EAMM_PASS_CHAINS=3 (22 + 21 + 21 frames).  Reference for the loop these calls replace: demo.py:251-281."""
import threading

import pytest
import torch

from conftest import TOL
from eamm_amd import OcclusionAwareGenerator, hot_path_config
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict
from oracle import eamm_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
KEYS = ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed")
# two launch plans of the same frames differ by summation order only (split-K slabs, transform-point groups per launch size)
PLAN_TOL = {"prediction": 2e-5, "mask": 5e-6, "occlusion_map": 5e-6, "sparse_deformed": 1e-4, "deformed": 1e-4}
_STATE = {}


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


def cuda(d):
    return {k: v.to(DEV) for k, v in d.items()}


def fresh_engine(max_frames, seed=1234):
    cfg = hot_path_config()
    gen = OcclusionAwareGenerator(**cfg, max_frames=max_frames)
    gen.load_state_dict(synthetic_state_dict(cfg, seed=seed), strict=True)
    gen = gen.to(DEV).eval()
    eng = gen.encode_source(synthetic_source(256, seed=1).to(DEV), max_frames=max_frames)
    return gen, eng


def state():
    """One 64-frame handle, one 16-frame handle and the 16-frame plan's result for 64 frames, shared by the tests."""
    if not _STATE:
        cfg = hot_path_config()
        _STATE["cfg"], _STATE["sd"] = cfg, synthetic_state_dict(cfg, seed=1234)
        _STATE["src"] = synthetic_source(256, seed=1)
        _STATE["kp_s"] = synthetic_keypoints(1, 10, seed=0)
        _STATE["kp_d"] = synthetic_keypoints(64, 10, seed=2)
        _STATE["g64"], _STATE["e64"] = fresh_engine(64)
        _STATE["g16"], _STATE["e16"] = fresh_engine(16)
        kd, ks = cuda(_STATE["kp_d"]), cuda(_STATE["kp_s"])
        parts = [_STATE["e16"].forward_frames({k: v[i:i + 16] for k, v in kd.items()}, ks, outputs=KEYS) for i in range(0, 64, 16)]
        _STATE["ref16"] = {k: torch.cat([p[k] for p in parts]) for k in KEYS}
        torch.cuda.synchronize()
    return _STATE


def worst(a, b):
    return {k: float((a[k] - b[k]).abs().max()) for k in KEYS}


def show(tag, errs):
    print("\n" + tag + "  " + "  ".join(f"{k}={v:.2e}" for k, v in errs.items()))


def test_64_frame_call_is_four_chains_and_matches_16_frame_calls_and_oracle():
    st = state()
    e64, e16 = st["e64"], st["e16"]
    assert e64.pass_chains(64) == 4 and e64.bottleneck_form(16) == 4 and e16.pass_chains(16) == 2
    plan = e64.describe_plan(64)
    assert plan["pass_chains"] == 4 and plan["frames_per_chain"] == 16 and plan["bottleneck_form"] == 4 and plan["side_streams"] == 3
    kd, ks = cuda(st["kp_d"]), cuda(st["kp_s"])
    out = e64.forward_frames(kd, ks, outputs=KEYS)
    e64.check_numeric()
    assert e64.last_stream_set() == 0          # an eager call of the only active thread runs on the device's shared pool
    errs = worst(out, st["ref16"])
    show("64 frames / four chains vs 4 x 16 frames / two chains", errs)
    for k in KEYS:
        assert errs[k] <= PLAN_TOL[k], (k, errs[k])
    frames = [0, 17, 40, 63]                   # one frame of each chain
    sel = {k: v[frames] for k, v in st["kp_d"].items()}
    ref = orc.generator_forward(st["sd"], st["cfg"], st["src"].expand(4, -1, -1, -1).contiguous(), sel,
                                {k: v.expand(4, *v.shape[1:]).contiguous() for k, v in st["kp_s"].items()})
    oerr = {k: float((out[k][frames].cpu() - ref[k]).abs().max()) for k in KEYS}
    show("64 frames / four chains vs oracle, frames 0 17 40 63", oerr)
    for k in KEYS:
        assert oerr[k] <= TOL[k], (k, oerr[k])
    # the same call again gives the same bits (no cross-chain hazard on the four workspace slabs)
    again = e64.forward_frames(kd, ks, outputs=KEYS)
    for k in KEYS:
        assert torch.equal(again[k], out[k]), k


@pytest.mark.parametrize("n,chains", [(50, 2), (33, 2), (63, 2), (17, 2)])
def test_ragged_calls_on_the_four_chain_handle(n, chains):
    """Below 64 frames the 64-frame handle falls back to two chains -- 25 + 25, 17 + 16, 32 + 31, 9 + 8 frames."""
    st = state()
    e64 = st["e64"]
    assert e64.pass_chains(n) == chains
    kd, ks = cuda(st["kp_d"]), cuda(st["kp_s"])
    out = e64.forward_frames({k: v[:n] for k, v in kd.items()}, ks, outputs=KEYS)
    assert out["prediction"].shape[0] == n
    errs = worst(out, {k: v[:n] for k, v in st["ref16"].items()})
    show(f"{n} frames on the 64-frame handle vs the 16-frame plan", errs)
    for k in KEYS:
        assert errs[k] <= PLAN_TOL[k], (n, k, errs[k])
    for t in (0, n - 1):                       # first frame of the first chain, last frame of the last
        ref = orc.generator_forward(st["sd"], st["cfg"], st["src"], {k: v[t:t + 1] for k, v in st["kp_d"].items()}, st["kp_s"])
        for k in KEYS:
            assert float((out[k][t].cpu() - ref[k][0]).abs().max()) <= TOL[k], (n, t, k)


def test_128_frame_call_four_chains_of_32():
    """The clip legs of bench.py batch 128 frames per call (four chains of 32 frames, each chain's bottleneck GEMM 512 workgroups):
    against eight 16-frame calls and against the oracle on the first and the last frame."""
    st = state()
    _, e128 = fresh_engine(128)
    assert e128.pass_chains(128) == 4 and e128.describe_plan(128)["frames_per_chain"] == 32
    kp_d = synthetic_keypoints(128, 10, seed=2)
    kd, ks = cuda(kp_d), cuda(st["kp_s"])
    out = e128.forward_frames(kd, ks, outputs=KEYS)
    parts = [st["e16"].forward_frames({k: v[i:i + 16] for k, v in kd.items()}, ks, outputs=KEYS) for i in range(0, 128, 16)]
    ref16 = {k: torch.cat([p[k] for p in parts]) for k in KEYS}
    errs = worst(out, ref16)
    show("128 frames / four chains of 32 vs 8 x 16 frames", errs)
    for k in KEYS:
        assert errs[k] <= PLAN_TOL[k], (k, errs[k])
    assert torch.equal(out["prediction"][:64], out["prediction"][:64]) and torch.equal(ref16["prediction"][:64], st["ref16"]["prediction"])
    for t in (0, 127):
        ref = orc.generator_forward(st["sd"], st["cfg"], st["src"], {k: v[t:t + 1] for k, v in kp_d.items()}, st["kp_s"])
        for k in KEYS:
            assert float((out[k][t].cpu() - ref[k][0]).abs().max()) <= TOL[k], (t, k)


def test_three_chains_by_knob(monkeypatch):
    """EAMM_PASS_CHAINS=3: 64 frames as 22 + 21 + 21."""
    st = state()
    monkeypatch.setenv("EAMM_PASS_CHAINS", "3")
    _, e3 = fresh_engine(64)
    assert e3.pass_chains(64) == 3
    out = e3.forward_frames(cuda(st["kp_d"]), cuda(st["kp_s"]), outputs=KEYS)
    errs = worst(out, st["ref16"])
    show("64 frames / three chains vs the 16-frame plan", errs)
    for k in KEYS:
        assert errs[k] <= PLAN_TOL[k], (k, errs[k])
    from eamm_amd.engine import library_knobs
    knobs = library_knobs()
    assert knobs["EAMM_PASS_CHAINS"] == {"value": 3, "set": 1}


def test_gemm_cache_policy_variants_are_bit_identical(monkeypatch):
    """The bottleneck GEMM's variant 6 = variant 3 with the transformed-input stream loaded non-temporally.  The library picks 6
    while every GEMM workgroup of a call has a CU of its own (16 frames at 256x256: + 0.3 %) and 3 for larger calls (128 frames:
    6 costs 2 %; eamm_api.hip wino4_variant_for); EAMM_WINO4_VARIANT pins one for every call size.  A cache hint must not change a
    bit: each plan under the other variant against the default's frames."""
    st = state()
    kd, ks = cuda(st["kp_d"]), cuda(st["kp_s"])
    assert st["e16"].describe_plan(16)["wino4_variant"] == 6 and st["e64"].describe_plan(16)["wino4_variant"] == 6
    assert st["e64"].describe_plan(17)["wino4_variant"] == 3 and st["e64"].describe_plan(64)["wino4_variant"] == 3
    ref64 = {k: v.clone() for k, v in st["e64"].forward_frames(kd, ks, outputs=KEYS).items()}
    monkeypatch.setenv("EAMM_WINO4_VARIANT", "6")
    _, e6 = fresh_engine(64)
    assert e6.describe_plan(64)["wino4_variant"] == 6
    out = e6.forward_frames(kd, ks, outputs=KEYS)
    for k in KEYS:
        assert torch.equal(out[k], ref64[k]), k
    monkeypatch.setenv("EAMM_WINO4_VARIANT", "3")
    _, e3 = fresh_engine(16)
    assert e3.describe_plan(16)["wino4_variant"] == 3
    part = e3.forward_frames({k: v[16:32] for k, v in kd.items()}, ks, outputs=KEYS)
    for k in KEYS:
        assert torch.equal(part[k], st["ref16"][k][16:32]), k


def test_two_handles_on_two_caller_streams_share_the_pool():
    """Two handles driven alternately from ONE host thread on two caller streams: their chains interleave on the pool's
    three side streams (shared order, separate fork / join events) -- the frames must be the serial run's, bit for bit."""
    st = state()
    ea = st["e64"]
    _, eb = fresh_engine(64)
    ks = cuda(st["kp_s"])
    kda, kdb = cuda(st["kp_d"]), cuda(synthetic_keypoints(64, 10, seed=500))
    want_a = ea.forward_frames(kda, ks)["prediction"].clone()
    want_b = eb.forward_frames(kdb, ks)["prediction"].clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    got_a, got_b = [], []
    for _ in range(3):
        with torch.cuda.stream(sa):
            got_a.append(ea.forward_frames(kda, ks)["prediction"])
        with torch.cuda.stream(sb):
            got_b.append(eb.forward_frames(kdb, ks)["prediction"])
    torch.cuda.synchronize()
    assert ea.last_stream_set() == 0 and eb.last_stream_set() == 0
    for g in got_a:
        assert torch.equal(g, want_a)
    for g in got_b:
        assert torch.equal(g, want_b)


def test_two_handles_from_two_threads():
    """Two host threads, each with its own handle and caller stream, enqueueing at the same time: whoever finds the pool
    leased runs that call on its private streams (eamm_last_stream_set 1).  Results == serial, bit for bit."""
    st = state()
    ea = st["e64"]
    _, eb = fresh_engine(64)
    ks = cuda(st["kp_s"])
    kds = [cuda(st["kp_d"]), cuda(synthetic_keypoints(64, 10, seed=500))]
    engines = [ea, eb]
    want = [e.forward_frames(kd, ks)["prediction"].clone() for e, kd in zip(engines, kds)]
    torch.cuda.synchronize()
    results, sets, errors = [[], []], [set(), set()], []
    barrier = threading.Barrier(2)

    def work(i):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            with torch.no_grad(), torch.cuda.stream(stream):
                barrier.wait()
                for _ in range(12):
                    results[i].append(engines[i].forward_frames(kds[i], ks)["prediction"])
                    sets[i].add(engines[i].last_stream_set())
            stream.synchronize()
        except Exception as exc:   # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    print(f"\nstream sets used by the two threads: {sets}")
    assert sets[0] | sets[1] <= {0, 1}
    for i in range(2):
        assert len(results[i]) == 12
        for g in results[i]:
            assert torch.equal(g, want[i]), i


def test_graph_capture_of_the_64_frame_call_uses_private_streams_while_another_handle_runs():
    """(d) The 64-frame call captured into a HIP graph: its three side streams are the HANDLE's (eamm_last_stream_set 2), so
    the capture never drags the device's shared pool in -- another handle keeps running eagerly on the pool from a second
    thread while the capture is open -- and the graph replays bit-exactly."""
    st = state()
    ea = st["e64"]
    _, eb = fresh_engine(64)
    ks = cuda(st["kp_s"])
    kda, kdb = cuda(st["kp_d"]), cuda(synthetic_keypoints(64, 10, seed=500))
    want_a = ea.forward_frames(kda, ks)["prediction"].clone()
    want_b = eb.forward_frames(kdb, ks)["prediction"].clone()
    torch.cuda.synchronize()
    got_b, errors = [], []
    go, stop = threading.Event(), threading.Event()

    def eager():
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            with torch.no_grad(), torch.cuda.stream(stream):
                got_b.append(eb.forward_frames(kdb, ks)["prediction"])      # allocator warm on this stream
                stream.synchronize()
                go.set()
                while not stop.is_set() and len(got_b) < 40:
                    got_b.append(eb.forward_frames(kdb, ks)["prediction"])
                    stream.synchronize()
        except Exception as exc:   # pragma: no cover
            errors.append(exc)
            go.set()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ea.forward_frames(kda, ks)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    th = threading.Thread(target=eager)
    th.start()
    go.wait(60)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        out = ea.forward_frames(kda, ks)["prediction"]
    assert ea.last_stream_set() == 2
    stop.set()
    th.join()
    assert not errors, errors
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want_a)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want_a)
    assert len(got_b) >= 2
    for g in got_b:
        assert torch.equal(g, want_b)
    # and an eager call after the capture is back on the pool
    ea.forward_frames(kda, ks)
    assert ea.last_stream_set() == 0
