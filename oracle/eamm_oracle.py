"""CPU oracle for the dense-motion + OcclusionAwareGenerator forward path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU restatement of the reference's
algorithm for the one hot path this repository accelerates.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product
path (``eamm_amd``) never does and fails loudly when its HIP library is missing.

Parity pin: the reference ships no tests, golden vectors or checkpoints for this path (SURVEY.md
section 4), so the oracle is pinned against the reference ITSELF: ``oracle/make_golden.py`` imports
``/root/reference`` (possible only in the build container), drives it with the seeded weights and
inputs of ``eamm_amd.weights`` and (a) asserts this restatement reproduces every output key, (b)
writes the reference's outputs as fixtures under ``tests/golden/``.  ``tests/test_oracle_golden.py``
re-checks the oracle against those fixtures wherever the tests run.

The restatement is functional (a state_dict in, tensors out) rather than a module tree; every
function cites the reference lines it follows.  dtype follows the inputs, so the same code gives
the fp64 noise floor of the algorithm.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # reference sync_batchnorm/batchnorm.py:39


# ----------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------
def coordinate_grid(h: int, w: int, dtype, device="cpu") -> torch.Tensor:
    """[h,w,2] grid, last dim (x, y), x = 2*j/(w-1)-1 -- reference util.py:839-855."""
    xs = 2 * (torch.arange(w, device=device).to(dtype) / (w - 1)) - 1
    ys = 2 * (torch.arange(h, device=device).to(dtype) / (h - 1)) - 1
    return torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], dim=2)


def gaussian_heatmaps(kp_value: torch.Tensor, h: int, w: int, variance: float) -> torch.Tensor:
    """[B,K,h,w] exp(-0.5*|z-kp|^2/var) -- reference util.py:815-836."""
    grid = coordinate_grid(h, w, kp_value.dtype, kp_value.device)          # [h,w,2]
    diff = grid[None, None] - kp_value[:, :, None, None, :]                # [B,K,h,w,2]
    return torch.exp(-0.5 * (diff ** 2).sum(-1) / variance)


_TRAIN = None   # generator_forward_train's context: {"parallel": bool, "stats": {prefix: (running_mean, running_var)}}


def batch_norm_eval(x, sd, prefix):
    """Eval-mode BatchNorm with running statistics -- sync_batchnorm/batchnorm.py:48-53.  Inside generator_forward_train the
    same call sites take the training branch instead (batch statistics, running statistics updated: batchnorm.py:55-125)."""
    if _TRAIN is not None:
        outs, rm, rv = sync_batchnorm_forward([x], sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"],
                                              sd[prefix + ".running_var"], parallel=_TRAIN["parallel"])
        _TRAIN["stats"][prefix] = (rm, rv)
        return outs[0]
    return F.batch_norm(x, sd[prefix + ".running_mean"].to(x.dtype), sd[prefix + ".running_var"].to(x.dtype),
                        sd[prefix + ".weight"].to(x.dtype), sd[prefix + ".bias"].to(x.dtype),
                        False, 0.0, BN_EPS)


def _conv(x, sd, prefix, pad):
    return F.conv2d(x, sd[prefix + ".weight"].to(x.dtype), sd[prefix + ".bias"].to(x.dtype), padding=pad)


def same_block(x, sd, prefix, pad):
    """conv -> BN -> ReLU -- reference util.py:934-938."""
    return F.relu(batch_norm_eval(_conv(x, sd, prefix + ".conv", pad), sd, prefix + ".norm"))


def down_block(x, sd, prefix):
    """conv3x3 -> BN -> ReLU -> avgpool 2x2 -- reference util.py:915-920."""
    return F.avg_pool2d(same_block(x, sd, prefix, 1), kernel_size=(2, 2))


def up_block(x, sd, prefix):
    """nearest x2 -> conv3x3 -> BN -> ReLU -- reference util.py:895-900."""
    return same_block(F.interpolate(x, scale_factor=2), sd, prefix, 1)


def res_block(x, sd, prefix):
    """x + conv2(relu(bn2(conv1(relu(bn1(x)))))) -- reference util.py:872-880."""
    y = _conv(F.relu(batch_norm_eval(x, sd, prefix + ".norm1")), sd, prefix + ".conv1", 1)
    y = _conv(F.relu(batch_norm_eval(y, sd, prefix + ".norm2")), sd, prefix + ".conv2", 1)
    return y + x


def hourglass(x, sd, prefix, num_blocks):
    """U-Net: encoder keeps every scale, decoder concatenates [up, skip] -- util.py:956-987."""
    skips = [x]
    for i in range(num_blocks):
        skips.append(down_block(skips[-1], sd, f"{prefix}.encoder.down_blocks.{i}"))
    out = skips.pop()
    for i in range(num_blocks):
        out = up_block(out, sd, f"{prefix}.decoder.up_blocks.{i}")
        out = torch.cat([out, skips.pop()], dim=1)
    return out


def antialias_down(x, sd, scale, key="dense_motion_network.down.weight"):
    """zero-pad 6, depthwise 13x13 Gaussian, keep every (1/scale)-th row/col -- util.py:1044-1052."""
    if scale == 1:
        return x
    wgt = sd[key].to(x.dtype)
    ka = wgt.shape[-1] // 2
    y = F.conv2d(F.pad(x, (ka, ka, ka, ka)), wgt, groups=x.shape[1])
    step = int(1 / scale)
    return y[:, :, ::step, ::step]


# ----------------------------------------------------------------------------------------------
# dense motion -- reference dense_motion.py:32-113
# ----------------------------------------------------------------------------------------------
def sparse_motions(kp_driving, kp_source, h, w):
    """[B,K+1,h,w,2]: T_0 = identity grid, T_k = J_s J_d^-1 (z - kp_d) + kp_s -- dense_motion.py:47-67."""
    vd, vs = kp_driving["value"], kp_source["value"]
    b, k = vd.shape[:2]
    grid = coordinate_grid(h, w, vs.dtype, vs.device)
    rel = grid[None, None] - vd[:, :, None, None, :]                          # [B,K,h,w,2]
    if "jacobian" in kp_driving:
        jac = torch.matmul(kp_source["jacobian"], torch.inverse(kp_driving["jacobian"]))  # [B,K,2,2]
        rel = torch.einsum("bkij,bkhwj->bkhwi", jac, rel)
    moved = rel + vs[:, :, None, None, :]
    ident = grid[None, None].expand(b, 1, h, w, 2)
    return torch.cat([ident, moved], dim=1)


def dense_motion(sd, cfg, source_image, kp_driving, kp_source):
    dm = cfg["dense_motion_params"]
    nk = cfg["num_kp"]
    var = dm.get("kp_variance", 0.01)
    src = antialias_down(source_image, sd, dm.get("scale_factor", 1))
    b, c, h, w = src.shape
    # heat-maps: [0, G(driving) - G(source)]                                 dense_motion.py:32-45
    heat = gaussian_heatmaps(kp_driving["value"], h, w, var) - gaussian_heatmaps(kp_source["value"], h, w, var)
    heat = torch.cat([torch.zeros_like(heat[:, :1]), heat], dim=1)[:, :, None]   # [B,K+1,1,h,w]
    motions = sparse_motions(kp_driving, kp_source, h, w)                       # [B,K+1,h,w,2]
    # K+1 backward warps of the small source                                dense_motion.py:69-79
    rep = src[:, None].expand(b, nk + 1, c, h, w).reshape(b * (nk + 1), c, h, w)
    warped = F.grid_sample(rep, motions.reshape(b * (nk + 1), h, w, 2), mode="bilinear",
                           padding_mode="zeros", align_corners=False)
    warped = warped.view(b, nk + 1, c, h, w)
    hg_in = torch.cat([heat, warped], dim=2).view(b, (nk + 1) * (c + 1), h, w)  # dense_motion.py:93-94
    feat = hourglass(hg_in, sd, "dense_motion_network.hourglass", dm["num_blocks"])
    mask = F.softmax(_conv(feat, sd, "dense_motion_network.mask", 3), dim=1)    # dense_motion.py:98-99
    deformation = (motions * mask[..., None]).sum(dim=1)                        # [B,h,w,2]  :101-104
    out = {"sparse_deformed": warped, "mask": mask, "deformation": deformation}
    if cfg.get("estimate_occlusion_map", False):
        out["occlusion_map"] = torch.sigmoid(_conv(feat, sd, "dense_motion_network.occlusion", 3))
    return out


# ----------------------------------------------------------------------------------------------
# generator -- reference generator.py:50-97
# ----------------------------------------------------------------------------------------------
def warp_by_flow(inp, deformation):
    """deform_input: resize the flow bilinearly if needed, then grid_sample -- generator.py:50-57."""
    h, w = inp.shape[2:]
    if deformation.shape[1] != h or deformation.shape[2] != w:
        deformation = F.interpolate(deformation.permute(0, 3, 1, 2), size=(h, w), mode="bilinear",
                                    align_corners=False).permute(0, 2, 3, 1)
    return F.grid_sample(inp, deformation, mode="bilinear", padding_mode="zeros", align_corners=False)


def encode_source(sd, cfg, source_image):
    """Frame-invariant encoder: first 7x7 block + down blocks -- generator.py:61-63."""
    out = same_block(source_image, sd, "first", 3)
    for i in range(cfg["num_down_blocks"]):
        out = down_block(out, sd, f"down_blocks.{i}")
    return out


def decode(sd, cfg, feat):
    """bottleneck res-blocks, up blocks, final 7x7 conv, sigmoid -- generator.py:89-93."""
    out = feat
    for i in range(cfg["num_bottleneck_blocks"]):
        out = res_block(out, sd, f"bottleneck.r{i}")
    for i in range(cfg["num_down_blocks"]):
        out = up_block(out, sd, f"up_blocks.{i}")
    return torch.sigmoid(_conv(out, sd, "final", 3))


def generator_forward(sd, cfg, source_image, kp_driving, kp_source):
    """Same contract as OcclusionAwareGenerator.forward (generator.py:59-97): dict of outputs."""
    feat = encode_source(sd, cfg, source_image)
    outputs = {}
    if cfg.get("dense_motion_params") is not None:
        dmo = dense_motion(sd, cfg, source_image, kp_driving, kp_source)
        outputs["mask"] = dmo["mask"]
        outputs["sparse_deformed"] = dmo["sparse_deformed"]
        outputs["deformation"] = dmo["deformation"]  # not returned by the reference; kept for tests
        feat = warp_by_flow(feat, dmo["deformation"])
        if "occlusion_map" in dmo:
            occ = dmo["occlusion_map"]
            outputs["occlusion_map"] = occ
            if occ.shape[2:] != feat.shape[2:]:
                occ = F.interpolate(occ, size=feat.shape[2:], mode="bilinear", align_corners=False)
            feat = feat * occ
        outputs["deformed"] = warp_by_flow(source_image, dmo["deformation"])
    outputs["prediction"] = decode(sd, cfg, feat)
    return outputs


def generator_forward_train(sd, cfg, source_image, kp_driving, kp_source, parallel=False):
    """OcclusionAwareGenerator.forward in .train() mode: every SynchronizedBatchNorm2d of the blocks (modules/util.py:858-938)
    normalises with the statistics of the batch.  `parallel` = the replicas' formula (inv_std = clamp(var, eps) ** -0.5,
    batchnorm.py:125) applied to the WHOLE batch -- what R replicas compute together, up to the order in which their float
    sums are added; False = F.batch_norm (one replica, batchnorm.py:48-53).
    Returns (outputs, {norm prefix: (new running_mean, new running_var)})."""
    global _TRAIN
    assert _TRAIN is None
    _TRAIN = {"parallel": parallel, "stats": {}}
    try:
        return generator_forward(sd, cfg, source_image, kp_driving, kp_source), _TRAIN["stats"]
    finally:
        _TRAIN = None


# ----------------------------------------------------------------------------------------------
# key-point detectors ("next" row N1) -- reference modules/keypoint_detector.py
# ----------------------------------------------------------------------------------------------
def kp_head(sd, cfg, feature_map):
    """7x7 (pad `pad`) conv -> spatial softmax / temperature -> soft-argmax value; jacobian conv -> heat-map
    weighted sum -- keypoint_detector.py:77-105 (KPDetector) and :180-205 (KPDetector_a), identical heads."""
    pad = cfg.get("pad", 0)
    k = cfg["num_kp"]
    pred = F.conv2d(feature_map, sd["kp.weight"].to(feature_map.dtype), sd["kp.bias"].to(feature_map.dtype), padding=pad)
    b, _, hh, ww = pred.shape
    heat = F.softmax(pred.view(b, k, -1) / cfg["temperature"], dim=2).view(b, k, hh, ww)
    grid = coordinate_grid(hh, ww, pred.dtype, pred.device)
    out = {"value": (heat[..., None] * grid[None, None]).sum(dim=(2, 3)), "heatmap": heat}
    if cfg.get("estimate_jacobian", False):
        njm = 1 if cfg.get("single_jacobian_map", False) else k
        jm = F.conv2d(feature_map, sd["jacobian.weight"].to(feature_map.dtype), sd["jacobian.bias"].to(feature_map.dtype),
                      padding=pad).reshape(b, njm, 4, hh, ww)
        out["jacobian"] = (heat[:, :, None] * jm).view(b, k, 4, -1).sum(dim=-1).view(b, k, 2, 2)
    return out


def kp_detector_forward(sd, cfg, x):
    """KPDetector.forward: anti-alias down, hourglass, head -- keypoint_detector.py:77-105."""
    x = antialias_down(x, sd, cfg.get("scale_factor", 1), key="down.weight")
    return kp_head(sd, cfg, hourglass(x, sd, "predictor", cfg["num_blocks"]))


def kp_detector_a_forward(sd, cfg, feature_map):
    """KPDetector_a.forward: the head only, on a given (audio-driven) feature map -- keypoint_detector.py:180-205."""
    return kp_head(sd, cfg, feature_map)


def deconv_tail(sd, x):
    """`AT_net2.decon` (reference util.py:559-576, evaluated per frame at :604-607): ConvTranspose2d(k6,s2,p1) on the
    [B,C,1,1] LSTM feature, then ConvTranspose2d(k4,s2,p1) layers, eval BatchNorm2d + ReLU after all but the last.
    `sd` uses the Sequential's keys ("0.weight", "1.running_mean", "3.weight", ...)."""
    n = 1 + max(int(k.split(".")[0]) for k in sd if k.endswith(".weight") and sd[k].dim() == 4) // 3
    out = x.reshape(x.shape[0], -1, 1, 1)
    for i in range(n):
        w, b = sd[f"{3 * i}.weight"].to(out.dtype), sd[f"{3 * i}.bias"].to(out.dtype)
        out = F.conv_transpose2d(out, w, b, stride=2, padding=1)
        if i + 1 < n:
            out = F.relu(batch_norm_eval(out, sd, f"{3 * i + 1}"))
    return out


def animate_clip(sd, cfg, source_image, kp_source, kp_driving_seq):
    """Counterpart of the per-frame loop in demo.py:251-281: one generator call per driving frame,
    prediction returned as float32 [H,W,3] arrays (np.transpose(pred, [0,2,3,1])[0])."""
    frames = []
    n = kp_driving_seq["value"].shape[0]
    with torch.no_grad():
        for t in range(n):
            kp_d = {k: v[t:t + 1] for k, v in kp_driving_seq.items()}
            out = generator_forward(sd, cfg, source_image, kp_d, kp_source)
            frames.append(out["prediction"][0].permute(1, 2, 0).contiguous().numpy())
    return frames


# ----------------------------------------------------------------------------------------------
# clip harness ("next" row N2) -- reference demo.py:112-132, 194-282, filter1.py:13-47
# ----------------------------------------------------------------------------------------------
class _LowPass:
    """filter1.py:13-26."""

    def __init__(self):
        self.prev_raw = None
        self.prev_filtered = None

    def process(self, value, alpha):
        s = value if self.prev_raw is None else alpha * value + (1.0 - alpha) * self.prev_filtered
        self.prev_raw, self.prev_filtered = value, s
        return s


class OneEuro:
    """filter1.py:29-47, statement for statement (tensor in, tensor out; one call per frame)."""

    def __init__(self, mincutoff=1.0, beta=0.0, dcutoff=1.0, freq=30):
        self.freq, self.mincutoff, self.beta, self.dcutoff = freq, mincutoff, beta, dcutoff
        self.x_filter, self.dx_filter = _LowPass(), _LowPass()

    def compute_alpha(self, cutoff):
        import numpy as np
        te = 1.0 / self.freq
        tau = 1.0 / (2 * np.pi * cutoff)
        return 1.0 / (1.0 + tau / te)

    def process(self, x):
        prev_x = self.x_filter.prev_raw
        dx = 0.0 if prev_x is None else (x - prev_x) * self.freq
        edx = self.dx_filter.process(dx, self.compute_alpha(self.dcutoff))
        cutoff = self.mincutoff + self.beta * abs(edx)
        return self.x_filter.process(x, self.compute_alpha(cutoff))


def smooth_sequence(seq, mincutoff, beta, dcutoff, freq, scale):
    """demo.py:241-250 (and :231-239 for the emotion displacements): `process(x * scale) / scale`, frame after frame."""
    f = OneEuro(mincutoff=mincutoff, beta=beta, dcutoff=dcutoff, freq=freq)
    return torch.stack([f.process(seq[t] * scale) / scale for t in range(seq.shape[0])])


def normalize_kp(kp_source, kp_driving, kp_driving_initial, adapt_movement_scale=False, use_relative_movement=False,
                 use_relative_jacobian=False):
    """demo.py:112-132 for ONE driving frame (the reference calls it per frame, demo.py:276)."""
    import numpy as np
    from scipy.spatial import ConvexHull
    scale = 1
    if adapt_movement_scale:
        source_area = ConvexHull(kp_source["value"][0].numpy()).volume
        driving_area = ConvexHull(kp_driving_initial["value"][0].numpy()).volume
        scale = np.sqrt(source_area) / np.sqrt(driving_area)
    new = dict(kp_driving)
    if use_relative_movement:
        diff = (kp_driving["value"] - kp_driving_initial["value"])
        diff = diff * scale
        new["value"] = diff + kp_source["value"]
        if use_relative_jacobian:
            jd = torch.matmul(kp_driving["jacobian"], torch.inverse(kp_driving_initial["jacobian"]))
            new["jacobian"] = torch.matmul(jd, kp_source["jacobian"])
    return new


def animation_keypoints(sd_kp, cfg_kp, sd_decon, sd_kpa, cfg_kpa, source_image, lstm_features, emo_driving=None,
                        relative=True, adapt_movement_scale=True):
    """The key-point side of make_animation_smooth (demo.py:194-277), per frame as the reference runs it:
    kp_source = kp_detector(source) (:206); kp_driving_initial = kp_detector_a(deco_out[:, 0]) (:207); per frame
    kp_detector_a(deco_out[:, t]) (:219) with deco_out[:, t] = decon(lstm_out[:, t]) (util.py:603-607); One-Euro smoothing of
    the emotion displacements (:231-239) and of the key points (:241-250); emotion offsets (:263-271); normalize_kp (:276).
    ``emo_driving``: the emotion network's per-frame output {'value': [T,E,2], 'jacobian': [T,E,2,2]} (that network is outside
    this path) or None.  Returns (kp_source, [normalised key points of frame t], raw key points, smoothed key points)."""
    with torch.no_grad():
        kp_source = {k: v for k, v in kp_detector_forward(sd_kp, cfg_kp, source_image).items() if k != "heatmap"}
        T = lstm_features.shape[0]
        frames = [kp_detector_a_forward(sd_kpa, cfg_kpa, deconv_tail(sd_decon, lstm_features[t:t + 1])) for t in range(T)]
        raw = {k: torch.cat([f[k] for f in frames]) for k in ("value", "jacobian")}
        kp_initial = {k: raw[k][:1] for k in raw}
        if emo_driving is not None:
            emo = {k: smooth_sequence(emo_driving[k], 1, 0.2, 1.0, 100, 100) for k in ("value", "jacobian")}
        smooth = {k: smooth_sequence(raw[k], 0.05, 8, 1.0, 100, 10) for k in raw}
        out = []
        for t in range(T):
            kp = {k: smooth[k][t:t + 1].clone() for k in smooth}
            if emo_driving is not None:                                  # opt.type == 'linear_3'
                for key in ("value", "jacobian"):
                    kp[key][:, 1] = kp[key][:, 1] + emo[key][t:t + 1][:, 0] * 0.2
                    kp[key][:, 4] = kp[key][:, 4] + emo[key][t:t + 1][:, 1]
                    kp[key][:, 6] = kp[key][:, 6] + emo[key][t:t + 1][:, 2]
            out.append(normalize_kp(kp_source, kp, kp_initial, adapt_movement_scale=adapt_movement_scale,
                                    use_relative_movement=relative, use_relative_jacobian=relative))
    return kp_source, out, raw, smooth



def sync_batchnorm_forward(shards, weight, bias, running_mean, running_var, eps=1e-5, momentum=0.1, training=True,
                           parallel=None):
    """`_SynchronizedBatchNorm.forward` (reference sync_batchnorm/batchnorm.py:46-125) for the list of per-replica
    inputs `shards` ([N_r,C,H,W] each).  Returns (outputs per replica, new running_mean, new running_var).

    * evaluation, or one replica that is not "parallel": F.batch_norm (batchnorm.py:48-53);
    * several replicas (`parallel`, default len(shards) > 1): per-replica sum / sum of squares (:61-64), added over the
      replicas in order (:102 ReduceAddCoalesced), `_compute_mean_std` (:110-125: mean, unbiased variance into the running
      statistics, inv_std = clamp(biased variance, eps) ** -0.5), then (x - mean) * (inv_std * weight) + bias (:74-79)."""
    parallel = len(shards) > 1 if parallel is None else parallel
    rm, rv = running_mean.clone(), running_var.clone()
    if not (parallel and training):
        assert len(shards) == 1
        out = F.batch_norm(shards[0], rm, rv, weight, bias, training, momentum, eps)   # updates rm, rv in place when training
        return [out], rm, rv
    c = shards[0].shape[1]
    flat = [x.reshape(x.shape[0], c, -1) for x in shards]
    size = sum(v.shape[0] * v.shape[2] for v in flat)
    sum_ = ssum = None
    for v in flat:                                                        # device order, as ReduceAddCoalesced adds them
        s1, s2 = v.sum(dim=0).sum(dim=-1), (v ** 2).sum(dim=0).sum(dim=-1)
        sum_, ssum = (s1, s2) if sum_ is None else (sum_ + s1, ssum + s2)
    assert size > 1, 'BatchNorm computes unbiased standard-deviation, which requires size > 1.'
    mean = sum_ / size
    sumvar = ssum - sum_ * mean
    unbias_var, bias_var = sumvar / (size - 1), sumvar / size
    rm = (1 - momentum) * rm + momentum * mean
    rv = (1 - momentum) * rv + momentum * unbias_var
    inv_std = bias_var.clamp(eps) ** -0.5
    outs = []
    for r, (x, v) in enumerate(zip(shards, flat)):
        if isinstance(weight, (list, tuple)):   # one parameter copy per replica (DataParallel's broadcast copies): their
            o = (v - mean[None, :, None]) * (inv_std * weight[r])[None, :, None] + bias[r][None, :, None]   # gradients stay apart
        elif weight is not None:
            o = (v - mean[None, :, None]) * (inv_std * weight)[None, :, None] + bias[None, :, None]
        else:
            o = (v - mean[None, :, None]) * inv_std[None, :, None]
        outs.append(o.view(x.shape))
    return outs, rm, rv
