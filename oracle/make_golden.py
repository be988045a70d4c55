"""Pin the oracle against the reference itself and write tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box):

    python oracle/make_golden.py

What it does
  1. imports the reference's OcclusionAwareGenerator read-only (an empty ``cv2`` stub module is put
     on sys.path because modules/util.py:6 imports cv2 for an unrelated helper);
  2. loads the seeded synthetic weights of ``eamm_amd.weights`` into it with a STRICT
     ``load_state_dict`` (the reference's checkpoint contract, demo.py:91);
  3. runs it on seeded inputs, asserts that ``oracle/eamm_oracle.py`` reproduces every output key,
     and records the fp32-vs-fp64 noise floor of the reference algorithm;
  4. stores the REFERENCE outputs (data only: inputs + expected outputs, full tensors for the tiny
     config, strided samples + statistics for the full-size ones) under tests/golden/.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from eamm_amd.config import hot_path_config, tiny_config  # noqa: E402
from eamm_amd.weights import synthetic_keypoints, synthetic_source, synthetic_state_dict  # noqa: E402
from oracle import eamm_oracle as orc  # noqa: E402

REFERENCE = "/root/reference"
GOLDEN = os.path.join(ROOT, "tests", "golden")
KEYS = ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed", "deformation")


def import_reference():
    stub = tempfile.mkdtemp(prefix="cv2stub_")
    open(os.path.join(stub, "cv2.py"), "w").close()
    sys.path[:0] = [stub, REFERENCE]
    from modules.generator import OcclusionAwareGenerator  # type: ignore
    return OcclusionAwareGenerator


def run_reference(gen, source, kp_d, kp_s):
    with torch.no_grad():
        out = gen(source, kp_source=kp_s, kp_driving=kp_d)
        # 'deformation' is internal to the reference (dense_motion.py:106); fetch it from the
        # sub-module so the flow itself is pinned too.
        out["deformation"] = gen.dense_motion_network(source_image=source, kp_driving=kp_d,
                                                      kp_source=kp_s)["deformation"]
    return {k: out[k] for k in KEYS}


def stats(t):
    t = t.double()
    return [float(t.mean()), float(t.std()), float(t.min()), float(t.max())]


def case(OAG, name, cfg, size, n_frames, seed_w, stride, with_jacobian=True, per_frame_source=False):
    sd = synthetic_state_dict(cfg, seed=seed_w)
    gen = OAG(**cfg).eval()
    gen.load_state_dict(sd, strict=True)
    nsrc = n_frames if per_frame_source else 1
    source = synthetic_source(size, seed=1, batch=nsrc, channels=cfg["num_channels"])
    kp_s = synthetic_keypoints(nsrc, cfg["num_kp"], seed=0)
    kp_d = synthetic_keypoints(n_frames, cfg["num_kp"], seed=2, jacobian=with_jacobian)
    if per_frame_source:      # the module contract: batch of independent (source, kp) pairs
        src_b, kp_s_b = source, kp_s
    else:                      # clip: one source, n_frames driving poses (demo.py:251-281)
        src_b = source.expand(n_frames, -1, -1, -1).contiguous()
        kp_s_b = {k: v.expand(n_frames, *v.shape[1:]).contiguous() for k, v in kp_s.items()}
    ref = run_reference(gen, src_b, kp_d, kp_s_b)
    with torch.no_grad():
        mine = orc.generator_forward(sd, cfg, src_b, kp_d, kp_s_b)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        ref64 = orc.generator_forward(sd64, cfg, src_b.double(), {k: v.double() for k, v in kp_d.items()},
                                      {k: v.double() for k, v in kp_s_b.items()})
    report = {}
    for k in KEYS:
        d = float((mine[k] - ref[k]).abs().max())
        floor = float((ref[k].double() - ref64[k]).abs().max())
        report[k] = {"oracle_vs_reference": d, "fp32_vs_fp64_floor": floor, "stats": stats(ref[k])}
        # the restatement calls the same ATen kernels in the same order: it must agree with the
        # reference to well inside the fp32 noise floor of the algorithm.
        assert d <= max(2e-6, 2 * floor), (name, k, d, floor)
    blob = {
        "source_seed": np.int64(1), "weight_seed": np.int64(seed_w), "size": np.int64(size),
        "stride": np.int64(stride), "per_frame_source": np.int64(per_frame_source),
        "kp_source_value": kp_s["value"].numpy(), "kp_source_jacobian": kp_s["jacobian"].numpy(),
        "kp_driving_value": kp_d["value"].numpy(),
    }
    if with_jacobian:
        blob["kp_driving_jacobian"] = kp_d["jacobian"].numpy()
    for k in KEYS:
        t = ref[k]
        s = stride if k in ("prediction", "deformed") else (max(1, stride // 4) if k == "sparse_deformed" else 1)
        if k == "deformation":
            blob[k] = t[:, ::s, ::s].contiguous().numpy()
        else:
            blob[k] = t[..., ::s, ::s].contiguous().numpy()
        blob[k + "_stats"] = np.asarray(report[k]["stats"], dtype=np.float64)
        blob[k + "_stride"] = np.int64(s)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **blob)
    return report


def adversarial_case(OAG, name, cfg, size, n_frames, stride, seed_w=1234):
    """the reference's outputs on a badly conditioned network -- eamm_amd.weights.adversarial_state_dict
    (BatchNorm variances 1e-4 .. 10, gains 0.25 .. 4 of either sign, dead channels, conv gain x 4) with running statistics
    calibrated to its own activations, a saturated 0 / 1 source and key points on the frame border.  The calibrated statistics
    travel in the fixture; the weights are rebuilt from the seed."""
    from eamm_amd.weights import adversarial_inputs, adversarial_state_dict
    source, kp_s, kp_d = adversarial_inputs(size, n_frames, cfg["num_kp"], cfg["num_channels"])
    src_b = source.expand(n_frames, -1, -1, -1).contiguous()
    kp_s_b = {k: v.expand(n_frames, *v.shape[1:]).contiguous() for k, v in kp_s.items()}
    # calibration: one batch-statistics pass (every BatchNorm normalises with the batch's own statistics, so the activations stay
    # in range whatever the gains); running statistics := those, variances perturbed by 10^U(-0.3, 0.3)
    sd0 = adversarial_state_dict(cfg, seed_w)
    with torch.no_grad():
        _, st = orc.generator_forward_train({k: (v.double() if v.is_floating_point() else v) for k, v in sd0.items()}, cfg,
                                            src_b.double(), {k: v.double() for k, v in kp_d.items()},
                                            {k: v.double() for k, v in kp_s_b.items()})
    rs = np.random.RandomState(seed_w + 99)
    bn = {}
    for norm, (rm, rv) in sorted(st.items()):      # momentum 0.1 from running (0, 1): batch mean = rm / 0.1, unbiased var = (rv - 0.9) / 0.1
        mean = (rm / 0.1).float().numpy()
        var = np.maximum(((rv - 0.9) / 0.1).float().numpy(), 1e-12) * (10.0 ** rs.uniform(-0.3, 0.3, rv.numel())).astype(np.float32)
        bn[norm] = (mean, var.astype(np.float32))
    sd = adversarial_state_dict(cfg, seed_w, bn_stats=bn)
    gen = OAG(**cfg).eval()
    gen.load_state_dict(sd, strict=True)
    ref = run_reference(gen, src_b, kp_d, kp_s_b)
    with torch.no_grad():
        mine = orc.generator_forward(sd, cfg, src_b, kp_d, kp_s_b)
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        ref64 = orc.generator_forward(sd64, cfg, src_b.double(), {k: v.double() for k, v in kp_d.items()},
                                      {k: v.double() for k, v in kp_s_b.items()})
    report = {}
    blob = {"weight_seed": np.int64(seed_w), "size": np.int64(size), "stride": np.int64(stride), "frames": np.int64(n_frames)}
    allvar = np.concatenate([v for _, v in bn.values()])
    blob["running_var_range"] = np.asarray([allvar.min(), np.median(allvar), allvar.max()], dtype=np.float64)
    for norm, (mean, var) in bn.items():
        blob["bn_mean/" + norm] = mean
        blob["bn_var/" + norm] = var
    for k in KEYS:
        d = float((mine[k] - ref[k]).abs().max())
        floor = float((ref[k].double() - ref64[k]).abs().max())
        report[k] = {"oracle_vs_reference": d, "fp32_vs_fp64_floor": floor, "stats": stats(ref[k])}
        assert torch.isfinite(ref[k]).all(), (name, k)
        assert d <= max(2e-6, 2 * floor), (name, k, d, floor)
        t = ref[k]
        s = stride if k in ("prediction", "deformed") else (max(1, stride // 4) if k == "sparse_deformed" else 1)
        blob[k] = (t[:, ::s, ::s] if k == "deformation" else t[..., ::s, ::s]).contiguous().numpy()
        blob[k + "_stride"] = np.int64(s)
        blob[k + "_floor"] = np.float64(floor)         # the reference's own fp32-vs-fp64 distance on this network
        blob[k + "_stats"] = np.asarray(report[k]["stats"], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **blob)
    return report


def no_motion_case(OAG):
    """The reference generator built WITHOUT a motion network (dense_motion_params=None, generator.py:18-23): forward is
    encoder -> bottleneck -> up blocks -> final, 'prediction' is the only output and the key points are never read."""
    cfg = tiny_config()
    cfg["dense_motion_params"] = None
    cfg["estimate_occlusion_map"] = False
    sd = synthetic_state_dict(cfg, seed=77)
    gen = OAG(**cfg).eval()
    gen.load_state_dict(sd, strict=True)
    src = synthetic_source(64, seed=3, batch=2)
    with torch.no_grad():
        ref = gen(src, kp_source=None, kp_driving=None)
        mine = orc.generator_forward(sd, cfg, src, None, None)
    assert sorted(ref) == ["prediction"] and sorted(mine) == ["prediction"]
    d = float((mine["prediction"] - ref["prediction"]).abs().max())
    assert d <= 2e-6, d
    np.savez_compressed(os.path.join(GOLDEN, "tiny64_nomotion.npz"), weight_seed=np.int64(77), source_seed=np.int64(3),
                        prediction=ref["prediction"].numpy())
    print("tiny64_nomotion: oracle-vs-reference", d, "keys", len(sd))


def reference_function(path, name, namespace):
    """Pull ONE function out of a reference file that cannot be imported as a module (demo.py loads dlib
    models at import time) and compile it as the reference wrote it."""
    import ast
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module(body=[fn], type_ignores=[]), path, "exec")
    exec(code, namespace)
    return namespace[name]


def normalize_kp_case():
    """Fixtures for the clip loop's key-point normalisation (demo.py:112-132), all flag combinations."""
    from scipy.spatial import ConvexHull
    ref = reference_function(os.path.join(REFERENCE, "demo.py"), "normalize_kp",
                             {"np": np, "torch": torch, "ConvexHull": ConvexHull})
    kp_s = synthetic_keypoints(1, 10, seed=0)
    kp_i = synthetic_keypoints(1, 10, seed=100)
    kp_d = synthetic_keypoints(5, 10, seed=2)
    blob = {"kp_source_value": kp_s["value"].numpy(), "kp_source_jacobian": kp_s["jacobian"].numpy(),
            "kp_initial_value": kp_i["value"].numpy(), "kp_initial_jacobian": kp_i["jacobian"].numpy(),
            "kp_driving_value": kp_d["value"].numpy(), "kp_driving_jacobian": kp_d["jacobian"].numpy()}
    for adapt in (0, 1):
        for rel in (0, 1):
            for relj in (0, 1):
                vals, jacs = [], []
                for t in range(5):   # the reference normalises one frame per call
                    one = {k: v[t:t + 1].clone() for k, v in kp_d.items()}
                    out = ref(kp_source={k: v.clone() for k, v in kp_s.items()}, kp_driving=one,
                              kp_driving_initial={k: v.clone() for k, v in kp_i.items()},
                              adapt_movement_scale=bool(adapt), use_relative_movement=bool(rel),
                              use_relative_jacobian=bool(relj))
                    vals.append(out["value"].numpy())
                    jacs.append(out["jacobian"].numpy())
                blob[f"value_a{adapt}r{rel}j{relj}"] = np.concatenate(vals)
                blob[f"jacobian_a{adapt}r{rel}j{relj}"] = np.concatenate(jacs)
    np.savez_compressed(os.path.join(GOLDEN, "normalize_kp.npz"), **blob)
    print("normalize_kp: wrote", len(blob), "arrays")


def reference_statements(path, func, pick):
    """Statements of a block inside a reference function that cannot be imported (see reference_function): `pick`
    selects the ast node whose body is wanted; the body is compiled as the reference wrote it."""
    import ast
    tree = ast.parse(open(path).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == func)
    node = next(n for n in ast.walk(fn) if pick(n))
    return compile(ast.Module(body=node.body, type_ignores=[]), path, "exec")


def emotion_case():
    """Fixture for the emotion key-point offsets of the clip loop (demo.py:263-271, `--add_emo`, type 'linear_3') and the
    `normalize_kp` call that follows (demo.py:276): the reference's own statements, run per frame as the loop runs them."""
    import ast
    from scipy.spatial import ConvexHull
    demo = os.path.join(REFERENCE, "demo.py")

    def is_offset_block(n):   # `if opt.type == 'linear_3':` whose body assigns into kp_driving[...][:, k]
        return (isinstance(n, ast.If) and isinstance(n.test, ast.Compare) and isinstance(n.test.left, ast.Attribute)
                and n.test.left.attr == "type" and isinstance(n.body[0], ast.Assign)
                and isinstance(n.body[0].targets[0], ast.Subscript))

    code = reference_statements(demo, "make_animation_smooth", is_offset_block)
    norm = reference_function(demo, "normalize_kp", {"np": np, "torch": torch, "ConvexHull": ConvexHull})
    t, k, e = 5, 10, 4
    kp_s = synthetic_keypoints(1, k, seed=0)
    kp_i = synthetic_keypoints(1, k, seed=100)
    kp_d = synthetic_keypoints(t, k, seed=2)
    rs = np.random.RandomState(33)
    emo = {"value": torch.from_numpy((0.05 * rs.standard_normal((t, e, 2))).astype(np.float32)),
           "jacobian": torch.from_numpy((0.05 * rs.standard_normal((t, e, 2, 2))).astype(np.float32))}
    blob = {"kp_source_value": kp_s["value"].numpy(), "kp_source_jacobian": kp_s["jacobian"].numpy(),
            "kp_initial_value": kp_i["value"].numpy(), "kp_initial_jacobian": kp_i["jacobian"].numpy(),
            "kp_driving_value": kp_d["value"].numpy(), "kp_driving_jacobian": kp_d["jacobian"].numpy(),
            "emo_value": emo["value"].numpy(), "emo_jacobian": emo["jacobian"].numpy()}
    off_v, off_j, nrm_v, nrm_j = [], [], [], []
    for f in range(t):
        ns = {"kp_driving": {key: v[f:f + 1].clone() for key, v in kp_d.items()},
              "emo_driving": {key: v[f:f + 1].clone() for key, v in emo.items()}}
        exec(code, ns)                                                         # demo.py:266-271
        off_v.append(ns["kp_driving"]["value"].numpy().copy())
        off_j.append(ns["kp_driving"]["jacobian"].numpy().copy())
        out = norm(kp_source={key: v.clone() for key, v in kp_s.items()}, kp_driving=ns["kp_driving"],
                   kp_driving_initial={key: v.clone() for key, v in kp_i.items()}, use_relative_movement=True,
                   use_relative_jacobian=True, adapt_movement_scale=True)       # demo.py:276
        nrm_v.append(out["value"].numpy())
        nrm_j.append(out["jacobian"].numpy())
    blob.update(offset_value=np.concatenate(off_v), offset_jacobian=np.concatenate(off_j),
                normalized_value=np.concatenate(nrm_v), normalized_jacobian=np.concatenate(nrm_j))
    assert np.abs(blob["offset_value"] - blob["kp_driving_value"]).max() > 1e-3   # the block did something
    np.savez_compressed(os.path.join(GOLDEN, "emotion_offsets.npz"), **blob)
    print("emotion_offsets: wrote", len(blob), "arrays")


def kp_detector_cases(only=None):
    """Fixtures for the key-point detectors (modules/keypoint_detector.py): the reference modules driven with
    seeded weights; the oracle must reproduce them; fp32-vs-fp64 noise floor recorded."""
    from eamm_amd.config import kp_detector_a_config, kp_detector_config, tiny_kp_config
    from eamm_amd.weights import kp_state_dict_spec
    import_reference()
    from modules.keypoint_detector import KPDetector, KPDetector_a  # type: ignore
    report = {}
    for name, cfg, size, audio in (("kp_tiny64", tiny_kp_config(), 64, False), ("kp_full256", kp_detector_config(), 256, False),
                                   ("kpa_tiny", tiny_kp_config(audio=True), 64, True),
                                   ("kpa_full", kp_detector_a_config(), 256, True),
                                   ("kp_tiny64_gray", {**tiny_kp_config(), "num_channels": 1}, 64, False),
                                   ("kp_tiny64_rgba", {**tiny_kp_config(), "num_channels": 4}, 64, False),
                                   ("kp_tiny64_six_channels", {**tiny_kp_config(), "num_channels": 6}, 64, False),
                                   # round 6: num_kp != 10 (keypoint_detector.py:24-41 accepts any)
                                   ("kp_tiny64_k1", {**tiny_kp_config(), "num_kp": 1}, 64, False),
                                   ("kp_tiny64_k5", {**tiny_kp_config(), "num_kp": 5}, 64, False),
                                   ("kp_tiny64_k15", {**tiny_kp_config(), "num_kp": 15}, 64, False),
                                   ("kp_tiny64_k30", {**tiny_kp_config(), "num_kp": 30}, 64, False),
                                   ("kpa_tiny_k1", {**tiny_kp_config(audio=True), "num_kp": 1}, 64, True),
                                   ("kpa_tiny_k5", {**tiny_kp_config(audio=True), "num_kp": 5}, 64, True),
                                   ("kpa_tiny_k15", {**tiny_kp_config(audio=True), "num_kp": 15}, 64, True),
                                   ("kpa_tiny_k30", {**tiny_kp_config(audio=True), "num_kp": 30}, 64, True)):
        if only and name not in only:
            continue
        sd = synthetic_state_dict(cfg, seed=77, spec=kp_state_dict_spec(cfg))
        mod = (KPDetector_a if audio else KPDetector)(**cfg).eval()
        mod.load_state_dict(sd, strict=True)
        b = 2
        if audio:   # feature map [B, block_expansion + num_channels_a, h, w] as AT_net2 produces it (35 x 64 x 64)
            rs = np.random.RandomState(5)
            hh = size // 4
            x = torch.from_numpy(rs.standard_normal((b, cfg["block_expansion"] + cfg["num_channels_a"], hh, hh)).astype(np.float32))
            fwd = orc.kp_detector_a_forward
        else:
            x = synthetic_source(size, seed=3, batch=b, channels=cfg["num_channels"])
            fwd = orc.kp_detector_forward
        with torch.no_grad():
            ref = mod(x)
            mine = fwd(sd, cfg, x)
            sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
            ref64 = fwd(sd64, cfg, x.double())
        blob = {"weight_seed": np.int64(77), "size": np.int64(size), "batch": np.int64(b)}
        if audio:
            blob["feature_map"] = x.numpy()
        report[name] = {}
        for k in ("value", "jacobian", "heatmap"):
            d = float((mine[k] - ref[k]).abs().max())
            floor = float((ref[k].double() - ref64[k]).abs().max())
            report[name][k] = {"oracle_vs_reference": d, "fp32_vs_fp64_floor": floor, "stats": stats(ref[k])}
            assert d <= max(2e-6, 2 * floor), (name, k, d, floor)
            blob[k] = ref[k].numpy()
            blob[k + "_floor"] = np.float64(floor)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **blob)
        print(name, {k: (f"{v['oracle_vs_reference']:.1e}", f"{v['fp32_vs_fp64_floor']:.1e}") for k, v in report[name].items()})
    path = os.path.join(GOLDEN, "summary_kp.json")
    if only and os.path.exists(path):   # adding a fixture without regenerating the others
        report = {**json.load(open(path)), **report}
    with open(path, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)


def smoothing_case():
    """Fixture for the key-point smoothing between the two loops of make_animation_smooth (demo.py:237-250): the
    reference's own filter1.OneEuroFilter, driven exactly as demo.py drives it (per frame, `.process(x*10)/10`, one
    filter object for values and one for jacobians), on a seeded random-walk key-point sequence."""
    import_reference()
    from filter1 import OneEuroFilter  # type: ignore
    rs = np.random.RandomState(21)
    t, k = 24, 10
    value = torch.from_numpy(np.cumsum(0.02 * rs.standard_normal((t, 1, k, 2)), 0).astype(np.float32))
    jac = torch.from_numpy((np.eye(2) + np.cumsum(0.03 * rs.standard_normal((t, 1, k, 2, 2)), 0)).astype(np.float32))
    blob = {"value": value[:, 0].numpy(), "jacobian": jac[:, 0].numpy()}
    for name, kw, sc in (("kp", dict(mincutoff=0.05, beta=8, dcutoff=1.0, freq=100), 10),       # demo.py:241-250
                         ("emo", dict(mincutoff=1, beta=0.2, dcutoff=1.0, freq=100), 100)):     # demo.py:231-239
        fv, fj = OneEuroFilter(**kw), OneEuroFilter(**kw)
        ov, oj = [], []
        for j in range(t):
            ov.append(fv.process(value[j].cpu() * sc) / sc)
            oj.append(fj.process(jac[j].cpu() * sc) / sc)
        blob[name + "_value"] = torch.cat(ov, 0).numpy()
        blob[name + "_jacobian"] = torch.cat(oj, 0).numpy()
    np.savez_compressed(os.path.join(GOLDEN, "one_euro.npz"), **blob)
    print("one_euro", {k: v.shape for k, v in blob.items()})


def deconv_tail_case():
    """Fixture for the deconvolution tail (N3): the reference's own `AT_net2().decon` (modules/util.py:559-576) with
    seeded weights, driven the way AT_net2.forward drives it -- one [1,256,1,1] LSTM feature per frame
    (util.py:600-607) -- plus the whole clip as one batch (what the drop-in does); the oracle must reproduce both."""
    from eamm_amd.weights import deconv_state_dict_spec
    import_reference()
    from modules.util import AT_net2  # type: ignore
    torch.manual_seed(0)
    net = AT_net2().eval()
    sd = synthetic_state_dict(None, seed=99, spec=deconv_state_dict_spec())
    net.decon.load_state_dict(sd, strict=True)
    t = 2
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.standard_normal((t, 256)).astype(np.float32))      # lstm_out[0, t, :]
    with torch.no_grad():
        per_frame = torch.cat([net.decon(x[i:i + 1, :, None, None]) for i in range(t)], 0)   # util.py:603-607
        batched = net.decon(x[:, :, None, None])
        mine = orc.deconv_tail(sd, x)
        ref64 = orc.deconv_tail({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double())
    d_batch = float((per_frame - batched).abs().max())
    d = float((mine - per_frame).abs().max())
    floor = float((per_frame.double() - ref64).abs().max())
    assert d <= max(2e-6, 2 * floor), (d, floor)
    blob = {"weight_seed": np.int64(99), "x": x.numpy(), "out": per_frame.numpy(), "floor": np.float64(floor)}
    np.savez_compressed(os.path.join(GOLDEN, "deconv_tail.npz"), **blob)
    rep = {"oracle_vs_reference": d, "fp32_vs_fp64_floor": floor, "per_frame_vs_batched_reference": d_batch,
           "stats": stats(per_frame)}
    with open(os.path.join(GOLDEN, "summary_deconv.json"), "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    print("deconv_tail", rep)


def batchnorm_case():
    """Fixture for the training-mode BatchNorm forward and backward (N4): the reference's own SynchronizedBatchNorm2d
    (sync_batchnorm/batchnorm.py) -- evaluation, training on one replica (F.batch_norm branch), and training on two
    replicas with UNEQUAL shards.  The two-replica case runs the reference's real protocol: a master and a slave copy
    registered through __data_parallel_replicate__, the two forwards on two threads talking through SyncMaster; only the
    two CUDA-only transport functions (ReduceAddCoalesced / Broadcast) are replaced by their CPU meaning (add / copy)."""
    import copy
    import threading
    import_reference()
    import sync_batchnorm.batchnorm as ref_bn  # type: ignore
    from sync_batchnorm.replicate import CallbackContext  # type: ignore

    def reduce_add(dest, n, *ts):   # sum of the per-device groups, in device order
        groups = [ts[i:i + n] for i in range(0, len(ts), n)]
        return tuple(sum(g[k] for g in groups[1:]) + groups[0][k] if len(groups) > 1 else groups[0][k] for k in range(n))

    ref_bn.ReduceAddCoalesced = type("ReduceAddCoalesced", (), {"apply": staticmethod(reduce_add)})
    ref_bn.Broadcast = type("Broadcast", (), {"apply": staticmethod(lambda gpus, *ts: tuple(t.clone() for _ in gpus for t in ts))})
    rs = np.random.RandomState(5)
    c, h, w = 40, 12, 16
    x = torch.from_numpy((rs.standard_normal((6, c, h, w)) * rs.uniform(0.5, 2.0, (1, c, 1, 1)) +
                          rs.standard_normal((1, c, 1, 1))).astype(np.float32))
    params = {"weight": torch.from_numpy(rs.uniform(0.5, 1.5, c).astype(np.float32)),
              "bias": torch.from_numpy((0.2 * rs.standard_normal(c)).astype(np.float32)),
              "running_mean": torch.from_numpy((0.3 * rs.standard_normal(c)).astype(np.float32)),
              "running_var": torch.from_numpy(rs.uniform(0.5, 2.0, c).astype(np.float32))}

    def fresh():
        m = ref_bn.SynchronizedBatchNorm2d(c)
        with torch.no_grad():
            m.weight.copy_(params["weight"]); m.bias.copy_(params["bias"])
            m.running_mean.copy_(params["running_mean"]); m.running_var.copy_(params["running_var"])
        return m

    blob = {"x": x.numpy(), **{k: v.numpy() for k, v in params.items()}, "split": np.int64(4)}
    with torch.no_grad():
        m = fresh().eval()
        blob["eval_out"] = m(x).numpy()
        m = fresh().train()
        blob["single_out"] = m(x).numpy()
        blob["single_running_mean"], blob["single_running_var"] = m.running_mean.numpy().copy(), m.running_var.numpy().copy()
        master = fresh().train()
        slave = copy.deepcopy(master)
        ctx = CallbackContext()
        master.__data_parallel_replicate__(ctx, 0)
        slave.__data_parallel_replicate__(ctx, 1)
        outs = [None, None]

        def run(i, mod, inp):
            with torch.no_grad():
                outs[i] = mod(inp)

        threads = [threading.Thread(target=run, args=(0, master, x[:4])), threading.Thread(target=run, args=(1, slave, x[4:]))]
        [t.start() for t in threads]
        [t.join() for t in threads]
        blob["sync_out"] = torch.cat(outs, 0).numpy()
        blob["sync_running_mean"], blob["sync_running_var"] = master.running_mean.numpy().copy(), master.running_var.numpy().copy()
        # the oracle must reproduce all three
        o, _, _ = orc.sync_batchnorm_forward([x], params["weight"], params["bias"], params["running_mean"], params["running_var"], training=False)
        assert float((o[0] - torch.from_numpy(blob["eval_out"])).abs().max()) <= 1e-6
        o, rm, rv = orc.sync_batchnorm_forward([x], params["weight"], params["bias"], params["running_mean"], params["running_var"])
        assert float((o[0] - torch.from_numpy(blob["single_out"])).abs().max()) <= 1e-6
        assert float((rv - torch.from_numpy(blob["single_running_var"])).abs().max()) <= 1e-6
        o, rm, rv = orc.sync_batchnorm_forward([x[:4], x[4:]], params["weight"], params["bias"], params["running_mean"], params["running_var"])
        d = float((torch.cat(o, 0) - torch.from_numpy(blob["sync_out"])).abs().max())
        assert d == 0.0 and torch.equal(rm, torch.from_numpy(blob["sync_running_mean"])) and torch.equal(rv, torch.from_numpy(blob["sync_running_var"])), d
    assert float(np.abs(blob["sync_out"] - blob["single_out"]).max()) < 1e-4   # same statistics, two formulas for inv_std
    # ---- backward (round 3): the reference modules under autograd with a fixed upstream gradient dy -- evaluation, one
    # replica, two replicas (the gradient crosses the replicas through the add / copy that stand for ReduceAddCoalesced /
    # Broadcast, exactly as it does through the CUDA originals); parameter gradients are each replica's own
    dy = torch.from_numpy(rs.standard_normal(x.shape).astype(np.float32))
    blob["dy"] = dy.numpy()

    def grads(m, xin, g):
        xin = xin.clone().requires_grad_(True)
        m.zero_grad()
        m(xin).backward(g)
        return xin.grad.numpy().copy(), m.weight.grad.numpy().copy(), m.bias.grad.numpy().copy()

    blob["eval_dx"], blob["eval_dw"], blob["eval_db"] = grads(fresh().eval(), x, dy)
    blob["single_dx"], blob["single_dw"], blob["single_db"] = grads(fresh().train(), x, dy)
    master = fresh().train()
    slave = copy.deepcopy(master)
    ctx = CallbackContext()
    master.__data_parallel_replicate__(ctx, 0)
    slave.__data_parallel_replicate__(ctx, 1)
    xs = [x[:4].clone().requires_grad_(True), x[4:].clone().requires_grad_(True)]
    outs = [None, None]

    def run_grad(i, mod, inp):
        with torch.enable_grad():
            outs[i] = mod(inp)

    threads = [threading.Thread(target=run_grad, args=(0, master, xs[0])), threading.Thread(target=run_grad, args=(1, slave, xs[1]))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    ((outs[0] * dy[:4]).sum() + (outs[1] * dy[4:]).sum()).backward()
    blob["sync_dx"] = torch.cat([xs[0].grad, xs[1].grad], 0).numpy()
    blob["sync_dw0"], blob["sync_db0"] = master.weight.grad.numpy().copy(), master.bias.grad.numpy().copy()
    blob["sync_dw1"], blob["sync_db1"] = slave.weight.grad.numpy().copy(), slave.bias.grad.numpy().copy()
    # the oracle under autograd must give the same gradients
    for key, shards, training in (("eval", [x], False), ("single", [x], True), ("sync", [x[:4], x[4:]], True)):
        ws = [params["weight"].clone().requires_grad_(True) for _ in shards]   # one parameter copy per replica, as above
        bs = [params["bias"].clone().requires_grad_(True) for _ in shards]
        xin = [t.clone().requires_grad_(True) for t in shards]
        if len(shards) == 1:
            o, _, _ = orc.sync_batchnorm_forward(xin, ws[0], bs[0], params["running_mean"], params["running_var"], training=training)
        else:
            o, _, _ = orc.sync_batchnorm_forward(xin, ws, bs, params["running_mean"], params["running_var"], training=training)
        (torch.cat(o, 0) * dy).sum().backward()
        d = float((torch.cat([t.grad for t in xin], 0) - torch.from_numpy(blob[key + "_dx"])).abs().max())
        assert d <= 2e-6, (key, d)
        if len(shards) == 1:
            assert float((ws[0].grad - torch.from_numpy(blob[key + "_dw"])).abs().max()) <= 2e-5
            assert float((bs[0].grad - torch.from_numpy(blob[key + "_db"])).abs().max()) <= 2e-5
        else:
            for r in (0, 1):
                assert float((ws[r].grad - torch.from_numpy(blob[f"sync_dw{r}"])).abs().max()) <= 2e-5
                assert float((bs[r].grad - torch.from_numpy(blob[f"sync_db{r}"])).abs().max()) <= 2e-5
    np.savez_compressed(os.path.join(GOLDEN, "batchnorm_train.npz"), **blob)
    print("batchnorm_train: wrote", {k: v.shape for k, v in blob.items() if hasattr(v, "shape")})


def train_mode_case(OAG):
    """Fixture for the generator's forward in .train() mode (N4, second slice): the reference generator with batch statistics
    in every SynchronizedBatchNorm2d, on ONE replica (F.batch_norm branch) and on TWO replicas with unequal shards (3 + 1
    pairs) driven through the reference's own replication callbacks and SyncMaster protocol on two threads (only the two
    CUDA-only transport functions are replaced by their CPU meaning, as in batchnorm_case).  Stored: every output key and
    the running statistics every BatchNorm ends up with (two replicas: the master's -- the slaves' are never updated)."""
    import copy
    import threading
    import sync_batchnorm.batchnorm as ref_bn  # type: ignore
    from sync_batchnorm.replicate import execute_replication_callbacks  # type: ignore

    def reduce_add(dest, n, *ts):
        groups = [ts[i:i + n] for i in range(0, len(ts), n)]
        return tuple(sum(g[k] for g in groups[1:]) + groups[0][k] if len(groups) > 1 else groups[0][k] for k in range(n))

    ref_bn.ReduceAddCoalesced = type("ReduceAddCoalesced", (), {"apply": staticmethod(reduce_add)})
    ref_bn.Broadcast = type("Broadcast", (), {"apply": staticmethod(lambda gpus, *ts: tuple(t.clone() for _ in gpus for t in ts))})
    cfg = tiny_config()
    sd = synthetic_state_dict(cfg, seed=1234)
    n, split = 4, 3
    source = synthetic_source(64, seed=1, batch=n)
    kp_s = synthetic_keypoints(n, cfg["num_kp"], seed=0)
    kp_d = synthetic_keypoints(n, cfg["num_kp"], seed=2)
    keys = ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed")
    norm_names = [k[:-len(".running_mean")] for k in sd if k.endswith(".running_mean")]

    def fresh():
        g = OAG(**cfg)
        g.load_state_dict(sd, strict=True)
        return g.train()

    blob = {"weight_seed": np.int64(1234), "n": np.int64(n), "split": np.int64(split), "norm_names": np.array(norm_names)}
    with torch.no_grad():
        g1 = fresh()
        out = g1(source, kp_source=kp_s, kp_driving=kp_d)
        st1 = g1.state_dict()
        for k in keys:
            blob["single_" + k] = out[k].numpy()
        for p in norm_names:
            blob["single_rm/" + p], blob["single_rv/" + p] = st1[p + ".running_mean"].numpy().copy(), st1[p + ".running_var"].numpy().copy()
        master = fresh()
        slave = copy.deepcopy(master)
        execute_replication_callbacks([master, slave])
        outs = [None, None]

        def run(i, mod, sl):
            with torch.no_grad():
                outs[i] = mod(source[sl], kp_source={k: v[sl] for k, v in kp_s.items()}, kp_driving={k: v[sl] for k, v in kp_d.items()})

        th = [threading.Thread(target=run, args=(0, master, slice(0, split))), threading.Thread(target=run, args=(1, slave, slice(split, n)))]
        [t.start() for t in th]
        [t.join() for t in th]
        stm = master.state_dict()
        for k in keys:
            blob["sync_" + k] = torch.cat([outs[0][k], outs[1][k]], 0).numpy()
        for p in norm_names:
            blob["sync_rm/" + p], blob["sync_rv/" + p] = stm[p + ".running_mean"].numpy().copy(), stm[p + ".running_var"].numpy().copy()
        assert all(torch.equal(slave.state_dict()[p + ".running_var"], sd[p + ".running_var"]) for p in norm_names)   # never updated
        # the oracle's training branch must reproduce both runs
        worst = {}
        for tag, par in (("single", False), ("sync", True)):
            mine, stats = orc.generator_forward_train(sd, cfg, source, kp_d, kp_s, parallel=par)
            assert sorted(stats) == sorted(norm_names)
            for k in keys:
                worst[tag + "_" + k] = float((mine[k] - torch.from_numpy(blob[tag + "_" + k])).abs().max())
            worst[tag + "_rm"] = max(float((stats[p][0] - torch.from_numpy(blob[tag + "_rm/" + p])).abs().max()) for p in norm_names)
            worst[tag + "_rv"] = max(float((stats[p][1] - torch.from_numpy(blob[tag + "_rv/" + p])).abs().max()) for p in norm_names)
        print("train_mode: oracle vs reference, worst |diff|:", {k: f"{v:.2e}" for k, v in worst.items()})
        # (bars: the evaluation fixtures' -- 'deformed' has a 1e-4 fp32 noise floor of its own)
        bar = {"prediction": 2e-5, "mask": 5e-6, "sparse_deformed": 1e-5, "occlusion_map": 5e-6, "deformed": 1e-4, "rm": 1e-6, "rv": 1e-6}
        assert all(v <= bar[k.split("_", 1)[1]] for k, v in worst.items()), worst
        d = float(np.abs(blob["single_prediction"] - blob["sync_prediction"]).max())
        assert d < 1e-3, d   # same statistics, two formulas for inv_std and two summation orders
    np.savez_compressed(os.path.join(GOLDEN, "tiny64_train.npz"), **blob)
    print("tiny64_train: wrote", len(blob), "arrays;", len(norm_names), "BatchNorm sites")


def train_backward_case(OAG, name="tiny64_train_backward", cfg=None, size=64, n=4, max_samples=8192, training=True):
    """Fixture for the generator's BACKWARD (N4): the reference generator on one replica, a scalar loss that
    weighs every output with fixed random tensors, ``loss.backward()`` -- the gradients autograd derives for every parameter,
    the key points (value + jacobian of driving and source) and the source image.  Stored in fp32 as the reference computes
    them, with the fp32-vs-fp64 distance of each (the same computation in double) so that tests can scale their bars to the
    algorithm's own noise floor.  The oracle must reproduce them.  ``tiny64_train_backward`` (.train(), 4 pairs at 64x64),
    ``full256_train_backward`` (.train(), the shipped configuration, 2 pairs at 256x256 -- the size at which the Winograd
    weight gradient, the thin 7x7 kernels on 64 channels and the split-group F(4x4) run inside the graph; round 4) and
    ``tiny64_eval_backward`` / ``full256_eval_backward`` (.eval(): running statistics -- the reference module is differentiable
    in evaluation mode too; without the batch statistics' cancellation the fp32 floor is 10-100x lower, so the 256x256 one is
    the SHARP check of the convolution / warp backward kernels at the shapes they run at)."""
    cfg = tiny_config() if cfg is None else cfg
    sd = synthetic_state_dict(cfg, seed=1234)
    source = synthetic_source(size, seed=1, batch=n)
    kp_s = synthetic_keypoints(n, cfg["num_kp"], seed=0)
    kp_d = synthetic_keypoints(n, cfg["num_kp"], seed=2)
    keys = ("prediction", "mask", "sparse_deformed", "occlusion_map", "deformed")
    gen = torch.Generator().manual_seed(77)
    full_outputs = size <= 64

    def run(dtype):
        g = OAG(**cfg)
        g.load_state_dict(sd, strict=True)
        g = g.to(dtype).train(training)
        src = source.detach().clone().to(dtype).requires_grad_()
        ks = {k: v.detach().clone().to(dtype).requires_grad_() for k, v in kp_s.items()}
        kd = {k: v.detach().clone().to(dtype).requires_grad_() for k, v in kp_d.items()}
        out = g(src, kp_source=ks, kp_driving=kd)
        loss = sum((out[k] * weights[k].to(dtype)).sum() for k in keys)
        loss.backward()
        grads = {"param/" + name: p.grad for name, p in g.named_parameters()}
        grads["source_image"] = src.grad
        for tag, kp in (("kp_source", ks), ("kp_driving", kd)):
            for k, v in kp.items():
                grads[tag + "/" + k] = v.grad
        missing = [k for k, v in grads.items() if v is None]
        assert not missing, (dtype, missing)
        return loss.detach(), {k: v.detach() for k, v in grads.items()}, {k: out[k].detach() for k in keys}

    with torch.no_grad():
        g0 = OAG(**cfg)
        g0.load_state_dict(sd, strict=True)
        shapes = {k: v.shape for k, v in g0.train(training)(source, kp_source=kp_s, kp_driving=kp_d).items() if k in keys}
    weights = {k: torch.randn(shapes[k], generator=gen) for k in keys}
    loss32, g32, out32 = run(torch.float32)
    loss64, g64, out64 = run(torch.float64)
    blob = {"weight_seed": np.int64(1234), "n": np.int64(n), "size": np.int64(size), "loss": np.float64(loss32),
            "loss64": np.float64(loss64), "names": np.array(sorted(g32)), "weights_seed": np.int64(77)}
    for k in keys:
        if full_outputs:
            blob["w/" + k] = weights[k].numpy()
            blob["out/" + k] = out32[k].numpy()
        else:   # the loss weights are torch.randn(shape, generator=manual_seed(77)) in `keys` order: regenerated by the test
            ostep = (out32[k].numel() // 16384) | 1
            blob["ostep/" + k] = np.int64(ostep)
            blob["out/" + k] = out32[k].reshape(-1)[::ostep].numpy()
            blob["out64/" + k] = out64[k].reshape(-1)[::ostep].float().numpy()
    floor = {}
    gmax = max(float(v.abs().max()) for v in g64.values())
    for k in sorted(g32):
        # large tensors: every step-th element of the flattened gradient (step odd, so no fixed phase against the 3x3 taps)
        step = 1 if g32[k].numel() <= max_samples else (g32[k].numel() // max_samples) | 1
        blob["step/" + k] = np.int64(step)
        blob["grad/" + k] = g32[k].reshape(-1)[::step].numpy() if step > 1 else g32[k].numpy()
        g64s = g64[k].reshape(-1)[::step] if step > 1 else g64[k]
        blob["grad64/" + k] = g64s.float().numpy()           # the same gradient computed in double (rounded for storage)
        scale = float(g64[k].abs().max())
        # a convolution bias in front of a BatchNorm has a gradient of exactly zero (the mean is subtracted again): what the
        # runs hold there is rounding noise, to be compared absolutely
        blob["zero/" + k] = np.bool_(scale < 1e-9 * gmax)
        blob["scale/" + k] = np.float64(scale)
        floor[k] = float((g32[k].double() - g64[k]).abs().max()) / (scale if scale >= 1e-9 * gmax else gmax)
        blob["floor/" + k] = np.float64(floor[k])
    nz = [k for k in floor if not blob["zero/" + k]]
    print(name + ": loss", float(loss32), "| largest gradient", f"{gmax:.3e}", "| fp32-vs-fp64 relative floor: worst",
          max(nz, key=floor.get), f"{max(floor[k] for k in nz):.2e}", "median", f"{float(np.median([floor[k] for k in nz])):.2e}",
          "|", len(floor) - len(nz), "identically-zero gradients")
    # the oracle's training branch, differentiated by autograd, must give the same gradients (in double, against the double run)
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and "running" not in k and "num_batches" not in k and "down.weight" not in k)
            for k, v in sd.items()}
    src = source.detach().double().requires_grad_()
    ks = {k: v.detach().double().requires_grad_() for k, v in kp_s.items()}
    kd = {k: v.detach().double().requires_grad_() for k, v in kp_d.items()}
    if training:
        mine, _ = orc.generator_forward_train(sd64, cfg, src, kd, ks, parallel=False)
    else:
        mine = orc.generator_forward(sd64, cfg, src, kd, ks)
    sum((mine[k] * weights[k].double()).sum() for k in keys).backward()
    worst = 0.0
    for k in sorted(g64):
        if k.startswith("param/"):
            got = sd64[k[len("param/"):]].grad
        elif k == "source_image":
            got = src.grad
        else:
            tag, key = k.split("/")
            got = (ks if tag == "kp_source" else kd)[key].grad
        assert got is not None, k
        rel = float((got - g64[k]).abs().max()) / (gmax if blob["zero/" + k] else float(g64[k].abs().max()))
        if rel > 1e-9:
            print('  oracle gradient differs:', k, rel, float(g64[k].abs().max()))
        worst = max(worst, rel)
    print(f"{name}: oracle autograd vs reference autograd (float64), worst relative |diff| {worst:.2e}")
    assert worst < 1e-9, worst
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **blob)
    print(name + ": wrote", len(blob), "arrays,", os.path.getsize(os.path.join(GOLDEN, name + ".npz")) >> 10, "KiB")


# num_kp != 10: (fixture, config, size, frames, sample stride, with driving jacobians).  K enters the 4(K+1)-channel hourglass input
# line, the softmax width of the flow head, its 7*(K+2)-column padding and the K + 2 <= 32 guard (eamm_api.hip): 1 = the smallest,
# 5 / 15 = below / above the shipped 10 (15: 4(K+1) = 64 fills the padded line exactly), 30 = the largest the library accepts
# (128 input channels, 7*32 = 224 flow-head columns); one case at the shipped 256x256 configuration, one without jacobians.
NUM_KP_CASES = (("tiny64_kp1", {**tiny_config(), "num_kp": 1}, 64, 2, 1, True),
                ("tiny64_kp5", {**tiny_config(), "num_kp": 5}, 64, 2, 1, True),
                ("tiny64_kp15", {**tiny_config(), "num_kp": 15}, 64, 2, 1, True),
                ("tiny64_kp30", {**tiny_config(), "num_kp": 30}, 64, 2, 1, True),
                ("tiny64_kp30_nojac", {**tiny_config(), "num_kp": 30}, 64, 2, 1, False),
                ("full256_kp15", {**hot_path_config(), "num_kp": 15}, 256, 2, 4, True),
                ("full256_kp5", {**hot_path_config(), "num_kp": 5}, 256, 1, 4, True))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "train_backward":
        os.makedirs(GOLDEN, exist_ok=True)
        train_backward_case(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "train_backward_256":
        torch.set_num_threads(os.cpu_count() or 1)
        train_backward_case(import_reference(), "full256_train_backward", hot_path_config(), 256, 2, max_samples=2048)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "eval_backward_256":
        torch.set_num_threads(os.cpu_count() or 1)
        train_backward_case(import_reference(), "full256_eval_backward", hot_path_config(), 256, 2, max_samples=2048, training=False)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "eval_backward":
        train_backward_case(import_reference(), "tiny64_eval_backward", tiny_config(), 64, 2, training=False)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "adversarial":   # badly conditioned statistics (added without regenerating the rest)
        torch.set_num_threads(os.cpu_count() or 1)
        OAG = import_reference()
        path = os.path.join(GOLDEN, "summary.json")
        summary = json.load(open(path))
        for name, cfg, size, n, stride in (("tiny64_adversarial", tiny_config(), 64, 3, 1), ("full256_adversarial", hot_path_config(), 256, 2, 4)):
            rep = adversarial_case(OAG, name, cfg, size, n, stride)
            summary["cases"][name] = rep
            print(name)
            for k, r in rep.items():
                print(f"   {k:16s} oracle-vs-ref {r['oracle_vs_reference']:.2e}  fp64 floor {r['fp32_vs_fp64_floor']:.2e}"
                      f"  mean/std/min/max {r['stats'][0]:.3f} {r['stats'][1]:.3f} {r['stats'][2]:.3f} {r['stats'][3]:.3f}")
        with open(path, "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "channels":   # num_channels 1 and 2 (generator.py:14,25,46; dense_motion.py:17-18,27)
        OAG = import_reference()
        which = sys.argv[2:] or ["tiny64_gray", "tiny64_two_channels", "tiny64_rgba", "tiny64_six_channels"]
        for name, ch in (("tiny64_gray", 1), ("tiny64_two_channels", 2), ("tiny64_rgba", 4), ("tiny64_six_channels", 6)):
            if name not in which:
                continue
            rep = case(OAG, name, {**tiny_config(), "num_channels": ch}, 64, 2, 1234, 1, per_frame_source=(ch in (2, 6)))
            print(name, {k: (r["oracle_vs_reference"], r["fp32_vs_fp64_floor"]) for k, r in rep.items()})
            path = os.path.join(GOLDEN, "summary.json")   # (added without regenerating the other fixtures)
            summary = json.load(open(path))
            summary["cases"][name] = rep
            with open(path, "w") as f:
                json.dump(summary, f, indent=1, sort_keys=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "num_kp":   # num_kp != 10 (dense_motion.py:15-18, generator.py:14 accept any)
        torch.set_num_threads(os.cpu_count() or 1)
        OAG = import_reference()
        path = os.path.join(GOLDEN, "summary.json")
        summary = json.load(open(path))
        which = sys.argv[2:]
        for name, cfg, size, n, stride, jac in NUM_KP_CASES:
            if which and name not in which:
                continue
            rep = case(OAG, name, cfg, size, n, 1234, stride, with_jacobian=jac)
            summary["cases"][name] = rep
            print(name, {k: (f"{r['oracle_vs_reference']:.1e}", f"{r['fp32_vs_fp64_floor']:.1e}") for k, r in rep.items()})
        with open(path, "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        kp_detector_cases(only=[f"{a}_k{k}" for a in ("kp_tiny64", "kpa_tiny") for k in (1, 5, 15, 30)])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "train":
        os.makedirs(GOLDEN, exist_ok=True)
        train_mode_case(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "nomotion":
        os.makedirs(GOLDEN, exist_ok=True)
        no_motion_case(import_reference())
        return
    if len(sys.argv) > 1 and sys.argv[1] == "batchnorm":
        os.makedirs(GOLDEN, exist_ok=True)
        batchnorm_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "smooth":
        os.makedirs(GOLDEN, exist_ok=True)
        smoothing_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "deconv":
        os.makedirs(GOLDEN, exist_ok=True)
        deconv_tail_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "normalize_kp":   # add this fixture without touching the others
        os.makedirs(GOLDEN, exist_ok=True)
        normalize_kp_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "emotion":
        os.makedirs(GOLDEN, exist_ok=True)
        emotion_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "kp":
        kp_detector_cases(only=sys.argv[2:] or None)
        return
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    os.makedirs(GOLDEN, exist_ok=True)
    OAG = import_reference()
    full, tiny = hot_path_config(), tiny_config()
    summary = {}
    summary["tiny64_clip3"] = case(OAG, "tiny64_clip3", tiny, 64, 3, 1234, 1)
    summary["tiny64_batch2"] = case(OAG, "tiny64_batch2", tiny, 64, 2, 1234, 1, per_frame_source=True)
    summary["tiny64_nojac"] = case(OAG, "tiny64_nojac", tiny, 64, 2, 1234, 1, with_jacobian=False)
    summary["full256_clip2"] = case(OAG, "full256_clip2", full, 256, 2, 1234, 4)
    summary["full512_clip1"] = case(OAG, "full512_clip1", full, 512, 1, 1234, 8)
    summary["tiny64_gray"] = case(OAG, "tiny64_gray", {**tiny, "num_channels": 1}, 64, 2, 1234, 1)
    summary["tiny64_two_channels"] = case(OAG, "tiny64_two_channels", {**tiny, "num_channels": 2}, 64, 2, 1234, 1, per_frame_source=True)
    summary["tiny64_rgba"] = case(OAG, "tiny64_rgba", {**tiny, "num_channels": 4}, 64, 2, 1234, 1)
    summary["tiny64_six_channels"] = case(OAG, "tiny64_six_channels", {**tiny, "num_channels": 6}, 64, 2, 1234, 1, per_frame_source=True)
    summary["tiny64_adversarial"] = adversarial_case(OAG, "tiny64_adversarial", tiny, 64, 3, 1)
    summary["full256_adversarial"] = adversarial_case(OAG, "full256_adversarial", full, 256, 2, 4)
    for name, cfg, size, n, stride, jac in NUM_KP_CASES:
        summary[name] = case(OAG, name, cfg, size, n, 1234, stride, with_jacobian=jac)
    no_motion_case(OAG)
    normalize_kp_case()
    emotion_case()
    kp_detector_cases()
    deconv_tail_case()
    smoothing_case()
    batchnorm_case()
    train_mode_case(OAG)
    train_backward_case(OAG)
    train_backward_case(OAG, "full256_train_backward", full, 256, 2, max_samples=2048)
    train_backward_case(OAG, "tiny64_eval_backward", tiny, 64, 2, training=False)
    train_backward_case(OAG, "full256_eval_backward", full, 256, 2, max_samples=2048, training=False)
    with open(os.path.join(GOLDEN, "summary.json"), "w") as f:
        json.dump({"torch": torch.__version__, "cases": summary}, f, indent=1, sort_keys=True)
    for name, rep in summary.items():
        print(name)
        for k, r in rep.items():
            print(f"   {k:16s} oracle-vs-ref {r['oracle_vs_reference']:.2e}  fp64 floor {r['fp32_vs_fp64_floor']:.2e}"
                  f"  mean/std/min/max {r['stats'][0]:.3f} {r['stats'][1]:.3f} {r['stats'][2]:.3f} {r['stats'][3]:.3f}")


if __name__ == "__main__":
    main()
